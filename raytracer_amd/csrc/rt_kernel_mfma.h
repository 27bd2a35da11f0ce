// Pipeline, part 1b: the broad phase on the matrix cores (broad_mfma_kernel, broad_mfma_binned_kernel).
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
//
// broad_kernel spends 6 of its 9 vector instructions per (ray, record) pair on two dot products, hb = a.c + od and
// nq = b.c + w.  A dot product of 32 rays with 32 records is a 32x32 matrix product, and v_mfma_f32_32x32x16_f16 does
// one with k = 16 in 32 cycles.  f16 keeps 11 significant bits, so every float is cut into two f16 pieces x = xh + xl
// (+ a remainder below 2^-21 |x|), and the product a c is taken as the four partial products
//     ah ch + ah cl + al ch + al cl        (dropped: the remainders, at most 2^-20 |a c|),
// each exact in the f32 accumulator.  Two MFMAs per 32 x 32 pairs (one for hb, one for N = thr - nq, k slots below) leave
// the vector ALU two instructions per pair -- t = N - hb^2 by one fma, its sign into the keep mask by one alignbit --
// instead of nine.  The result is used exactly like broad_kernel's: a conservative filter in front of the exact test.
//
// Range: f16 ends at 65504 and has a fixed absolute resolution of 2^-24 below 2^-14 (subnormals, which the conversions
// and the MFMA keep: measured).  All lengths are therefore scaled by a power of two s (exact in float) chosen from the
// scene so that the largest record coordinate lands near 64: squares stay below 2^12, low pieces stay far above 2^-24.
// broad_setup hands the ray over with its origin moved to the point of its line nearest the scene's centre (at most two
// scene radii out), so the scaled constants of a finite ray stay in range; a ray whose constants leave it all the same is
// not "ordinary": its list is marked as overflowed and the exact test sees every solid.
//
// Error bounds (u = 2^-24, S = |o| + cmax as in broad_setup): hb drops at most 2^-20 |a||c| + 2^-21 |od| <= 24 u S and
// rounds at most 16 additions of terms bounded by S: below Eh = 64 u S.  nq drops at most 2^-20 * 2|o||c| <= 8 u S^2 and
// rounds 16 additions: below E0 = 96 u S^2.  thr and w have no room for a third piece; their two pieces are rounded in
// the direction that keeps more (thr down, w up), which widens the test by at most 2^-21 of each and drops nothing.
//
// k slots (lane half h = lane >> 5 supplies k = 8h .. 8h+7 of both operands; c = record centre, w = its fourth word,
// a = -d', od, nb = -2o, thr = the ray constants of broad_setup; everything scaled):
//   record fragment F:   h=0 [ch 1 | cl 1]         h=1 [ch wh | cl wl]
//   hb = F x q0,  q0:    h=0 [ah odh | ah odl]     h=1 [al 0  | al 0 ]
//   N  = F x q1,  q1:    h=0 [nbh thh | nbh thl]   h=1 [nbl -1 | nbl -1]
// Output tile: column = ray (lane & 31), row = (reg & 3) + 8 (reg >> 2) + 4 h.  Record 16 h' + reg of a group sits in the
// row that lane half h' reads in register reg, so a lane's 16 sign bits are the keep bits of 16 consecutive records.
#pragma once

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int MF_BLOCK = 512;                 // threads per workgroup of broad_mfma_kernel (256 to 1024, 1 or 2 tiles per wave: the same time)
constexpr int MF_GROUP_BYTES = 1024;          // one fragment x 64 lanes x 16 bytes per group of 32 records
constexpr int MF_GROUP_WORDS = MF_GROUP_BYTES / 16;
constexpr int MF_WINDOW_GROUPS = 64;          // most groups per LDS window (64 KiB: two workgroups per CU)
static_assert(MF_WINDOW_GROUPS <= 64, "a tile keeps its window's group mask in two 32-bit words");
constexpr float MF_RANGE = 60000.0f;          // scaled ray constants must stay below this (f16 ends at 65504)

// ---- host side: the pieces of the records -------------------------------------------------------
// x cut to the f16 grid towards zero (11 significant bits; multiples of 2^-24 below 2^-14): exact float arithmetic
inline float f16_trunc(float x) {
    union { float f; unsigned u; } a;
    a.f = x;
    const unsigned e = (a.u >> 23) & 0xffu;
    if (e == 0xffu) return x;
    if (e >= 127u - 14u) { a.u &= 0xffffe000u; return a.f; }
    return truncf(x * 16777216.0f) * (1.0f / 16777216.0f);
}
// bit pattern of a float that lies on the f16 grid (or is infinite)
inline unsigned f16_bits(float v) {
    union { float f; unsigned u; } a;
    a.f = v;
    const unsigned sign = (a.u >> 16) & 0x8000u, e = (a.u >> 23) & 0xffu, man = a.u & 0x7fffffu;
    if (e == 0xffu) return sign | 0x7c00u | (man ? 0x200u : 0u);
    if (e == 0u) return sign;
    const int E = (int)e - 127;
    if (E > 15) return sign | 0x7c00u;
    if (E >= -14) return sign | ((unsigned)(E + 15) << 10) | (man >> 13);
    const int shift = -14 - E;
    return shift > 11 ? sign : sign | (((0x800000u | man) >> 13) >> shift);
}
inline float f16_value(unsigned b) {
    const int e = (int)((b >> 10) & 0x1fu);
    const float m = (float)(b & 0x3ffu);
    float v = e == 0 ? ldexpf(m, -24) : (e == 31 ? INFINITY : ldexpf(1024.0f + m, e - 25));
    return (b & 0x8000u) ? -v : v;
}
// two f16 pieces of x, cut towards zero; up = true: the second piece is raised by one step if the sum fell short of x,
// so that h + l >= x (h + l - x below 2^-21 |x|)
inline void f16_split2(float x, bool up, unsigned& h, unsigned& l) {
    const float xh = f16_trunc(x);
    h = f16_bits(xh);
    if (!(fabsf(x) <= 3.0e38f)) { l = 0; return; }     // infinite (a padding record's fourth word) or NaN: the first piece says it all
    const float r = x - xh;                            // exact
    float xl = f16_trunc(r);
    l = f16_bits(xl);
    if (up && xl < r) {
        l = (l & 0x8000u) ? (l == 0x8000u ? 0x0001u : l - 1u) : l + 1u;      // next value of the f16 grid above xl
        xl = f16_value(l);
    }
}

// the power of two by which lengths are scaled: the largest record extent (|c| + R) lands in (32, 64]
inline float mf_scale_for(float cmax) {
    if (!(cmax > 0.0f) || !(cmax < 3.0e38f)) return 1.0f;
    int e;
    frexpf(cmax, &e);                                  // cmax = m 2^e, m in [0.5, 1)
    int k = 6 - e;
    if (k > 100) k = 100;
    if (k < -100) k = -100;
    return ldexpf(1.0f, k);
}

__host__ __device__ inline uint4 pack8(unsigned e0, unsigned e1, unsigned e2, unsigned e3, unsigned e4, unsigned e5, unsigned e6, unsigned e7) {
    return make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), e4 | (e5 << 16), e6 | (e7 << 16));
}
constexpr unsigned F16_ONE = 0x3c00u, F16_MINUS_ONE = 0xbc00u;

// the row of the output tile (= row of the record operand) in which record k of a group is placed
__host__ __device__ inline int mf_row_of_record(int k) { const int hp = k >> 4, reg = k & 15; return (reg & 3) + 8 * (reg >> 2) + 4 * hp; }

// the fragment of lane `lane` for one group of 32 records (rec: 32 x float4 in stream order), lengths scaled by s
inline uint4 mf_record_frag(const float* rec, int lane, float s) {
    const int r = lane & 31, h = lane >> 5;
    int k = 0;
    for (int q = 0; q < 32; q++) if (mf_row_of_record(q) == r) k = q;
    const float* c = rec + 4 * k;
    unsigned ch[3], cl[3], wh, wl;
    for (int i = 0; i < 3; i++) f16_split2(c[i] * s, false, ch[i], cl[i]);
    if (fabsf(c[3]) <= 3.0e38f) {
        f16_split2(c[3] * s * s, true, wh, wl);        // w rounded up: a larger w keeps more
    } else {
        // a record that is never kept (w = -inf: padding, solids tested for every ray elsewhere) or always kept
        // (w = +inf: unbounded) carries the centre 0; the largest finite f16 decides N = thr - w against every ordinary
        // ray's |thr| < MF_RANGE, and unlike an infinity it can meet a zero in the hb product's k slot
        wh = c[3] < 0.0f ? 0xfbffu : 0x7bffu;
        wl = 0u;
    }
    if (h == 0) return pack8(ch[0], ch[1], ch[2], F16_ONE, cl[0], cl[1], cl[2], F16_ONE);
    return pack8(ch[0], ch[1], ch[2], wh, cl[0], cl[1], cl[2], wl);
}

// ---- device side ------------------------------------------------------------------------------
union F16Pack { uint4 u; f16x8 v; };
struct RayFrag { f16x8 q0, q1; };

typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2;
DEV unsigned mf_pk(float x, float y) {       // (f16(x), f16(y)) cut towards zero, x in the low half
    union { fp16x2 h; unsigned u; } p;
    p.h = __builtin_amdgcn_cvt_pkrtz(x, y);
    return p.u;
}
DEV float mf_lo(unsigned p) { union { unsigned u; fp16x2 h; } q; q.u = p; return (float)q.h[0]; }
DEV float mf_hi(unsigned p) { union { unsigned u; fp16x2 h; } q; q.u = p; return (float)q.h[1]; }

// scaled ray constants inside the f16 range?  (a ray record of broad_setup that is not of ordinary size carries thr = -inf)
DEV bool mf_ordinary(const float4 ra, const float4 rb, float s) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(rb.x), __builtin_fabsf(rb.y)), __builtin_fmaxf(__builtin_fabsf(rb.z), __builtin_fabsf(ra.w)));
    return (rb.w > -__builtin_inff()) & (__builtin_fabsf(rb.w) * s * s < MF_RANGE) & (m * s < MF_RANGE);
}

// the ray operands of the two MFMAs for this lane's half h; a ray that is not ordinary gets zeros (its list is marked
// as overflowed by the caller, so the exact test sees every solid)
DEV RayFrag mf_ray_frag(const float4 ra, const float4 rb, float s, int h, bool ordinary) {
    const float od = ra.w * s, n0 = -rb.x * s, n1 = -rb.y * s, n2 = -rb.z * s;
    float thr = rb.w * s * s;
    if (thr < 0.0f) thr = __builtin_fmaf(thr, 0x1p-20f, thr);      // cut towards zero below: make sure the pieces sum to <= thr
    // first pieces, their remainders (exact), second pieces
    const unsigned A01 = mf_pk(ra.x, ra.y), A2o = mf_pk(ra.z, od), N01 = mf_pk(n0, n1), N2t = mf_pk(n2, thr);
    const unsigned a01 = mf_pk(ra.x - mf_lo(A01), ra.y - mf_hi(A01)), a2o = mf_pk(ra.z - mf_lo(A2o), od - mf_hi(A2o));
    const unsigned m01 = mf_pk(n0 - mf_lo(N01), n1 - mf_hi(N01)), m2t = mf_pk(n2 - mf_lo(N2t), thr - mf_hi(N2t));
    F16Pack p0, p1;
    if (h == 0) {
        p0.u = make_uint4(A01, A2o, A01, (A2o & 0xffffu) | (a2o & 0xffff0000u));
        p1.u = make_uint4(N01, N2t, N01, (N2t & 0xffffu) | (m2t & 0xffff0000u));
    } else {
        const unsigned al2 = a2o & 0xffffu, nl2 = (m2t & 0xffffu) | (F16_MINUS_ONE << 16);
        p0.u = make_uint4(a01, al2, a01, al2);
        p1.u = make_uint4(m01, nl2, m01, nl2);
    }
    if (!ordinary) { p0.u = make_uint4(0u, 0u, 0u, 0u); p1.u = p0.u; }
    RayFrag f;
    f.q0 = p0.v; f.q1 = p1.v;
    return f;
}

// mf_ray_frag for both lane halves at once: out[0..1] = (q0, q1) of half 0, out[2..3] = those of half 1
DEV void mf_ray_frag_both(const float4 ra, const float4 rb, float s, bool ordinary, uint4* out) {
    const float od = ra.w * s, n0 = -rb.x * s, n1 = -rb.y * s, n2 = -rb.z * s;
    float thr = rb.w * s * s;
    if (thr < 0.0f) thr = __builtin_fmaf(thr, 0x1p-20f, thr);
    const unsigned A01 = mf_pk(ra.x, ra.y), A2o = mf_pk(ra.z, od), N01 = mf_pk(n0, n1), N2t = mf_pk(n2, thr);
    const unsigned a01 = mf_pk(ra.x - mf_lo(A01), ra.y - mf_hi(A01)), a2o = mf_pk(ra.z - mf_lo(A2o), od - mf_hi(A2o));
    const unsigned m01 = mf_pk(n0 - mf_lo(N01), n1 - mf_hi(N01)), m2t = mf_pk(n2 - mf_lo(N2t), thr - mf_hi(N2t));
    const unsigned al2 = a2o & 0xffffu, nl2 = (m2t & 0xffffu) | (F16_MINUS_ONE << 16);
    out[0] = make_uint4(A01, A2o, A01, (A2o & 0xffffu) | (a2o & 0xffff0000u));
    out[1] = make_uint4(N01, N2t, N01, (N2t & 0xffffu) | (m2t & 0xffff0000u));
    out[2] = make_uint4(a01, al2, a01, al2);
    out[3] = make_uint4(m01, nl2, m01, nl2);
    if (!ordinary) { out[0] = make_uint4(0u, 0u, 0u, 0u); out[1] = out[0]; out[2] = out[0]; out[3] = out[0]; }
}
// For every lane: the value of the lane of the LOWER half that sits 32 below it (lanes of the lower half get their own)
DEV unsigned mf_from_lower_half(unsigned v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[0]; }

// the two products of one tile (32 rays) with one group (32 records): hb and N = thr - nq, 16 values each per lane
DEV void mf_products(const uint4 fu, const RayFrag& rf, f32x16& hb, f32x16& n) {
    F16Pack f;
    f.u = fu;
    const f32x16 zero = {0};
    hb = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v, rf.q0, zero, 0, 0, 0);
    n = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v, rf.q1, zero, 0, 0, 0);
}

// keep bits of this lane's ray against the 16 records its lane half sees: bit 15 - reg = record 16 h + reg of the group
DEV unsigned mf_signs(const f32x16& hb, const f32x16& n) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) m = __builtin_amdgcn_alignbit(m, __float_as_uint(__builtin_fmaf(-hb[i], hb[i], n[i])), 31u);
    return m;
}

// mf_signs for group records with the behind test of broad_setup: a group whose bounding sphere lies wholly behind the
// ray's origin (hbt = hb - tmv >= 0 and hbt^2 > (s - thr) + mr; never with an infinite tmv; the callers replace a NaN by that) is not kept
DEV unsigned mf_signs_behind(const f32x16& hb, const f32x16& n, float tmv, float mr) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        // On sign bits, no compare and no select: kept = tt < 0 and not (hbt >= 0 and mr - (hbt^2 + tt) < 0).  A difference
        // that is zero comes out as +0 (not behind); hbt = +0 counts as in front, which the derivation allows (B >= 0 and
        // B^2 > disc leave no root at or past the origin either).
        const float tt = __builtin_fmaf(-hb[i], hb[i], n[i]);
        const float hbt = hb[i] - tmv;
        const float y = mr - __builtin_fmaf(hbt, hbt, tt);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(tt) & (__float_as_uint(hbt) | ~__float_as_uint(y)), 31u);
    }
    return m;
}

// The ray record at queue position pos, or `none` for a lane without a ray: the lane then reads position 0 (a queue that
// is scanned has one) and drops what it read -- a read under the condition is cut into one-word reads behind a branch each.
DEV void mf_load_ray(const float4* __restrict__ rec0, const float4* __restrict__ rec1, unsigned pos, bool valid, const float4 none, float4& ra, float4& rb) {
    const unsigned p = valid ? pos : 0u;
    const float4 a = rec0[p], b = rec1[p];
    ra = valid ? a : none; rb = valid ? b : none;
}

constexpr int MF_TILES = 2;                   // ray tiles (of 32 rays) per wave and batch

// For a lane of the lower half: the value of its partner lane (lane + 32).  v_permlane32_swap exchanges the upper half
// of its first operand with the lower half of its second, so with both operands = v the second comes back holding the
// upper half's values in the lower lanes.  (Lanes of the upper half get their own value back.)
DEV unsigned mf_from_upper_half(unsigned v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1]; }

// One workgroup's share of a window: batches of BLOCK/64 x MF_TILES x 32 rays against the groups [g_begin, g_end) staged
// in LDS.  SLICED: several workgroups scan different groups for the same rays and append through the list's counter;
// otherwise a ray's list is this lane's alone and its length stays in a register (no memory wait inside the loop).
// OR over the 64 lanes of a wave (DPP within rows of 16, then the four rows)
DEV unsigned mf_wave_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);   // row_mirror
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

// Inclusive sums over the lanes of each half of the wave (lane l receives v of lanes 32 (l / 32) .. l): four shifted adds
// within the rows of 16 lanes, then the lower row's total into the upper row -- data-parallel primitives, no LDS trip.
DEV unsigned mf_prefix32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    return v;
}

// CULL: the groups' own bounding spheres ("group records", gn_window of them in fragments that follow the window's
// groups in LDS) are tested first, and a group is scanned for a tile only if some ray of the tile keeps its group
// record.  Worth its two extra passes per tile only while the rays of a tile run side by side: the camera rays' round.
template <int BLOCK, bool SLICED, bool CULL>
DEV void mf_scan_window(const WfParams& w, const uint4* mf_lds, float scale, unsigned* cand_cnt, unsigned count, unsigned nbatches,
                        unsigned batch0, unsigned batch_step, int g0, int g_begin, int g_end, bool first_window, int gn_window,
                        unsigned& tiles_done) {
    constexpr unsigned per_block = (BLOCK / 64) * MF_TILES * 32;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
    const size_t cstride = w.slots;
    for (unsigned batch = batch0; batch < nbatches; batch += batch_step) {
        RayFrag rf[MF_TILES];
        unsigned pos[MF_TILES], n[MF_TILES];
        uint2* out[MF_TILES];                         // where this ray's next list entry goes (unsliced scan)
        bool own[MF_TILES];
#pragma unroll
        for (int t = 0; t < MF_TILES; t++) {
            pos[t] = batch * per_block + (unsigned)(wv * MF_TILES + t) * 32u + (unsigned)r;
            const bool valid = pos[t] < count;
            const float4 none = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
            float4 ra, rb;
            mf_load_ray(w.rec[0], w.rec[1], pos[t], valid, none, ra, rb);
            const bool ordinary = mf_ordinary(ra, rb, scale);
            rf[t] = mf_ray_frag(ra, rb, scale, h, ordinary);
            own[t] = valid && h == 0;                 // the lower lane half appends for the ray
            n[t] = 0;
            if (own[t]) {
                if (!ordinary) { if (SLICED || first_window) cand_cnt[pos[t]] = WF_OVERFLOW; }
                else if (!first_window) { n[t] = cand_cnt[pos[t]]; if (n[t] > (unsigned)WF_CAND_CAP) n[t] = (unsigned)WF_CAND_CAP + 1u; }
            }
            if (!ordinary) own[t] = false;
            out[t] = w.cand + ((size_t)n[t] * cstride + pos[t]);      // not used by a lane that does not own a ray
        }
        // one (tile, group) result: the two lane halves' bits joined, then the entry appended to the ray's list
        auto consume = [&](int t, int g, const f32x16& hb, const f32x16& nn) {
            const unsigned mine = mf_signs(hb, nn);
            const unsigned other = mf_from_upper_half(mine);
            const unsigned full = (mine << 16) | other;       // meaningful in the lower lane half
            if (full != 0u && own[t]) {
                const unsigned j = (unsigned)(g0 + g);
                if (SLICED) {
                    const unsigned at = atomicAdd(&cand_cnt[pos[t]], 1u);
                    if (at < (unsigned)WF_CAND_CAP) w.cand[(size_t)at * cstride + pos[t]] = make_uint2(j, full);
                } else {
                    // past the cap the list only counts on; such a count is written out as overflow
                    if (n[t] < (unsigned)WF_CAND_CAP) *out[t] = make_uint2(j, full);
                    out[t] += cstride;
                    n[t]++;                            // at most one per group of the stream on top of CAP + 1: no wrap
                }
            }
        };
        // which groups of the window each tile has to scan: bit 31 - (g & 31) of word g >> 5 (at most 64 groups a window)
        unsigned need[MF_TILES][2];
#pragma unroll
        for (int t = 0; t < MF_TILES; t++) { need[t][0] = 0xffffffffu; need[t][1] = 0xffffffffu; }
        if (CULL) {
            for (int l = 0; l * BROAD_GROUP < gn_window; l++) {
                const uint4 f = mf_lds[(gn_window + l) * MF_GROUP_WORDS + lane];
#pragma unroll
                for (int t = 0; t < MF_TILES; t++) {
                    f32x16 hb, nn;
                    mf_products(f, rf[t], hb, nn);
                    const unsigned mine = mf_signs(hb, nn);
                    const unsigned full = (mine << 16) | mf_from_upper_half(mine);
                    // (a ray that is not ordinary has its list marked as overflowed and asks for nothing here)
                    const unsigned any = mf_wave_or(own[t] ? full : 0u);
                    if (l == 0) need[t][0] = any; else need[t][1] = any;
                    if (CULL) tiles_done++;
                }
            }
        }
        if (!CULL) tiles_done += (unsigned)(g_end - g_begin) * MF_TILES;      // every group for every tile
        for (int g = g_begin; g < g_end; g++) {
            const unsigned bit = 0x80000000u >> (g & 31);
            bool want[MF_TILES], any_want = false;
#pragma unroll
            for (int t = 0; t < MF_TILES; t++) { want[t] = ((g < 32 ? need[t][0] : need[t][1]) & bit) != 0u; any_want |= want[t]; }
            if (!any_want) continue;
            const uint4 f = mf_lds[g * MF_GROUP_WORDS + lane];
#pragma unroll
            for (int t = 0; t < MF_TILES; t++) {
                if (!want[t]) continue;
                f32x16 hb, nn;
                mf_products(f, rf[t], hb, nn);
                consume(t, g, hb, nn);
                if (CULL) tiles_done++;
            }
        }
        if (!SLICED) {
#pragma unroll
            for (int t = 0; t < MF_TILES; t++) if (own[t]) cand_cnt[pos[t]] = n[t] > (unsigned)WF_CAND_CAP ? WF_OVERFLOW : n[t];
        }
    }
}

// Same contract as broad_kernel: for every queued ray the non-zero 32-record keep masks, appended to its list (an entry
// names a group of the table's own record order, see mf_build_table).  frags: per window its groups' fragments followed
// by the fragments of its group records; CULL: test the group records first (the camera rays' round).
template <int BLOCK, bool CULL>
__global__ __launch_bounds__(BLOCK, 6) void broad_mfma_kernel(const WfParams w, const uint4* __restrict__ frags, int n_groups, int win_groups, float scale) {
    extern __shared__ uint4 mf_lds[];
    const unsigned count = w.counts[w.iter];
    if (count == 0) return;
    unsigned* const cand_cnt = w.cand_cnt + (size_t)(w.iter & 1) * w.slots;
    constexpr unsigned per_block = (BLOCK / 64) * MF_TILES * 32;
    const unsigned nbatches = (count + per_block - 1) / per_block;
    const int n_windows = (n_groups + win_groups - 1) / win_groups;
    const int win_words = (win_groups + (win_groups + BROAD_GROUP - 1) / BROAD_GROUP) * MF_GROUP_WORDS;
    // fewer batches than workgroups: the groups of the (single) window are shared out in slices, see broad_kernel
    unsigned slices = 1;
    if (n_windows == 1 && nbatches < gridDim.x) {
        slices = gridDim.x / nbatches;
        if (slices > (unsigned)n_groups) slices = (unsigned)n_groups;
    }
    if (blockIdx.x >= nbatches * slices) return;
    // (tile, group) products of this lane's wave (the same in all its lanes); summed per workgroup: one atomic per wave
    // on the one counter is a queue of microseconds at the end of a short launch
    unsigned tiles_done = 0;
    __shared__ unsigned tiles_wg;
    if (threadIdx.x == 0) tiles_wg = 0u;
    for (int win = 0; win < n_windows; win++) {
        const int g0 = win * win_groups;
        const int gn = n_groups - g0 < win_groups ? n_groups - g0 : win_groups;
        const int staged = (gn + (gn + BROAD_GROUP - 1) / BROAD_GROUP) * MF_GROUP_WORDS;
        if (win > 0) __syncthreads();
        for (int i = threadIdx.x; i < staged; i += BLOCK) mf_lds[i] = frags[(size_t)win * win_words + i];
        __syncthreads();
        if (slices > 1) {
            const unsigned slice = blockIdx.x / nbatches;
            mf_scan_window<BLOCK, true, false>(w, mf_lds, scale, cand_cnt, count, nbatches, blockIdx.x % nbatches, nbatches, g0,
                                               (int)((unsigned)gn * slice / slices), (int)((unsigned)gn * (slice + 1) / slices), true, gn, tiles_done);
        } else if (CULL) {
            mf_scan_window<BLOCK, false, true>(w, mf_lds, scale, cand_cnt, count, nbatches, blockIdx.x, gridDim.x, g0, 0, gn, win == 0, gn, tiles_done);
        } else {
            mf_scan_window<BLOCK, false, false>(w, mf_lds, scale, cand_cnt, count, nbatches, blockIdx.x, gridDim.x, g0, 0, gn, win == 0, gn, tiles_done);
        }
    }
    {
        if ((threadIdx.x & 63) == 0) atomicAdd(&tiles_wg, tiles_done);
        __syncthreads();
        if (threadIdx.x == 0 && tiles_wg) atomicAdd((unsigned long long*)(w.kp.ctrl + 14), (unsigned long long)tiles_wg);
    }
}

// ---- the scan for rays that do not run side by side: binned by group -----------------------------------------------
// A ray's line touches the bounding spheres of few groups (C3: 5 of 32), but the 32 rays of a tile of a later round
// point every way and together touch them all, so asking the group records per tile drops nothing there.  This kernel
// asks them per RAY and then turns the work round: a workgroup takes 32 rays per wave (512 with 16 waves), every ray is
// tested against the window's group records and entered in the LDS list of each group it keeps; then the waves take
// (group, 32 listed rays) items in turn -- the rays' fragments come back out of LDS by the list -- and only
// these products are computed.  A kept word goes to its ray's list through a per-ray counter in LDS; the lists' order
// is that of arrival (as with the sliced scan).  Needs windows of at most 32 groups (one group-record fragment).
constexpr int MFB_GROUPS = 32;                 // groups per window for this kernel
constexpr int MFB_LIST = 192;                  // places in a group's ray list; a group more rays than that keep is scanned for all
                                               // (list_cap <= MFB_LIST: the number in use, lower in tests of that path)

#ifndef MFB_KARG
#define MFB_KARG 1          // 0: the parameters as the compiler keeps them (7 spilled VGPRs)
#endif
#ifndef MFB_WRITELANE
#define MFB_WRITELANE 1     // 0: the ballot words reach their lanes through compare + select
#endif
#ifndef MFB_TRANSPOSE
#define MFB_TRANSPOSE 1     // 0: the (ray, group) bit matrix is turned round by 32 ballots, each written into its two lanes
#endif
// The 32 x 32 bit matrix held by each half of the wave (lane r, bit b), transposed: lane k of the half receives in bit r
// what lane r held in bit k.  Five exchanges of off-diagonal blocks (16, 8, 4, 2, 1): a lane takes its partner's word
// (lane ^ j, through the LDS crossbar: no memory is touched), rotates it by j towards the bits it is to fill and inserts
// those bits -- three instructions per step.
template <int J, int LOG2J>
DEV unsigned mf_transpose_step(unsigned x, int lane) {
    constexpr unsigned M = J == 16 ? 0x0000ffffu : J == 8 ? 0x00ff00ffu : J == 4 ? 0x0f0f0f0fu : J == 2 ? 0x33333333u : 0x55555555u;   // bits b with (b & J) == 0
    const unsigned y = (unsigned)__builtin_amdgcn_ds_swizzle((int)x, (J << 10) | 0x1f);      // lane ^ J
    const unsigned u = (unsigned)__builtin_amdgcn_sbfe(lane, LOG2J, 1);                      // all ones in the lanes with (lane & J) != 0
    // those lanes fill the bits with (b & J) == 0 from y's bit b + J, the others the bits with (b & J) != 0 from y's bit b - J
    const unsigned rot = __builtin_amdgcn_alignbit(y, y, (u & (unsigned)J) | (~u & (unsigned)(32 - J)));
    const unsigned keep = M ^ u;
    return (x & keep) | (rot & ~keep);
}
DEV unsigned mf_transpose32(unsigned x, int lane) {
    asm volatile("" : "+v"(lane));        // the lanes' masks are worked out here, not kept in registers by a caller's loop
    x = mf_transpose_step<16, 4>(x, lane);
    x = mf_transpose_step<8, 3>(x, lane);
    x = mf_transpose_step<4, 2>(x, lane);
    x = mf_transpose_step<2, 1>(x, lane);
    return mf_transpose_step<1, 0>(x, lane);
}
#ifndef MFB_WIDE
#define MFB_WIDE 1          // 0: every wave prepares 32 rays, both lane halves building the same ray's pieces (scan +0.11 ms per C3 frame)
#endif
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 8) void broad_mfma_binned_kernel(const WfParams w_arg, const uint4* __restrict__ frags, int n_groups, int win_groups, float scale, int list_cap, int behind) {
    extern __shared__ uint4 mf_lds[];
    const WfParams& w = w_arg;
    constexpr int NW = BLOCK / 64, RAYS = NW * 32;
    constexpr int NBUILD = NW / 2;                         // MFB_WIDE: the waves that prepare a pass's rays, 64 each
    constexpr unsigned SKIP = 0x80000000u;                 // a list length from here on: not a list (the appends count on from it and never store)
    const unsigned count = w.counts[w.iter];
    if (count == 0) return;
    unsigned* const cand_cnt = w.cand_cnt + (size_t)(w.iter & 1) * w.slots;
    const unsigned nbatches = (count + RAYS - 1) / RAYS;
    if (blockIdx.x >= nbatches) return;
    const int n_windows = (n_groups + win_groups - 1) / win_groups;
    const int win_words = (win_groups + 1) * MF_GROUP_WORDS;
    // LDS: the window's fragments (groups, then the group records), the rays' fragments [ray][half][q0, q1], the groups'
    // ray lists, the rays' list lengths, the groups' list lengths (two copies, by pass parity)
    uint4* const gfr = mf_lds;
    uint4* const rayf = gfr + win_words;
    unsigned short* const lists = (unsigned short*)(rayf + (RAYS + 1) * 4);     // fragment RAYS: zeros, for the places of an item no ray fills
    unsigned* const rcnt = (unsigned*)(lists + MFB_GROUPS * MFB_LIST);
    unsigned* const gcount2 = rcnt + RAYS;
    // (the wave's number as a scalar: what follows from it -- item numbers, their groups and chunks -- then stays out of the vector unit)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), h = lane >> 5, r = lane & 31;
    unsigned pass = 0, tiles_done = 0;                     // (32 rays x 32 records) products of this wave
    for (int win = 0; win < n_windows; win++) {
        const int g0 = win * win_groups;
        const int gn = n_groups - g0 < win_groups ? n_groups - g0 : win_groups;
        __syncthreads();                                       // the previous window's last items are done
        for (int i = threadIdx.x; i < (gn + 1) * MF_GROUP_WORDS; i += BLOCK) gfr[i] = frags[(size_t)win * win_words + i];
        if (threadIdx.x < 2 * MFB_GROUPS) gcount2[threadIdx.x] = 0u;
        if (threadIdx.x < 4) rayf[RAYS * 4 + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (unsigned batch = blockIdx.x; batch < nbatches; batch += gridDim.x, pass++) {
            unsigned* const gcount = gcount2 + (pass & 1u) * MFB_GROUPS;
#if MFB_WIDE
            // ---- the first half of the waves: 64 rays each -- a lane builds both lane halves' fragments of its ray (the
            // pieces are the same work for both), the group records are asked in two tiles of 32 rays whose operands come
            // back out of LDS, one 64-bit ballot per group serves all 64 rays, and the lanes k and k + 32 enter the wave's
            // lower and upper 32 rays of group k in its list
            const unsigned local = (unsigned)(wv * 64 + lane);
            const unsigned pos = batch * RAYS + local;
            const bool builder = wv < NBUILD;
            const bool valid = builder && pos < count;
            if (builder) {
                const float4 none = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
#if MFB_KARG
                const WfParams& wa = karg_view();
#else
                const WfParams& wa = w_arg;
#endif
                float4 ra, rb;
                mf_load_ray(wa.rec[0], wa.rec[1], pos, valid, none, ra, rb);
                float tmv = __builtin_inff(), mr = 0.0f;      // the behind test's constants, in the table's scaled lengths
                if (valid && behind) { const float2 t = wa.tm[pos]; tmv = t.x * scale; mr = t.y * scale * scale; }
                if (!(tmv == tmv) || !(mr == mr)) { tmv = __builtin_inff(); mr = 0.0f; }      // (the sign-bit form of the test has no NaN case of its own)
                const bool ordinary = mf_ordinary(ra, rb, scale);
                uint4 fr4[4];
                mf_ray_frag_both(ra, rb, scale, ordinary, fr4);
#pragma unroll
                for (int k = 0; k < 4; k++) rayf[local * 4 + k] = fr4[k];
                const bool own = valid && ordinary;
                {
                    unsigned n0 = SKIP;                        // no ray here, or one whose list is marked as overflowed
                    if (valid && !ordinary) { if (win == 0) cand_cnt[pos] = WF_OVERFLOW; }
                    else if (valid) {
                        n0 = 0u;
                        if (win > 0) { n0 = cand_cnt[pos]; if (n0 > (unsigned)WF_CAND_CAP) n0 = (unsigned)WF_CAND_CAP + 1u; }
                    }
                    rcnt[local] = n0;
                }
                // the constants of the tile's rays: tile 0's sit in the lower lane half, tile 1's in the upper
                const unsigned tmv_u = __float_as_uint(tmv), mr_u = __float_as_uint(mr);
                const float tmv0 = __uint_as_float(mf_from_lower_half(tmv_u)), tmv1 = __uint_as_float(__builtin_amdgcn_permlane32_swap(tmv_u, tmv_u, false, false)[1]);
                const float mr0 = __uint_as_float(mf_from_lower_half(mr_u)), mr1 = __uint_as_float(__builtin_amdgcn_permlane32_swap(mr_u, mr_u, false, false)[1]);
                const uint4 gf = gfr[gn * MF_GROUP_WORDS + lane];
                unsigned full0, full1;
                {
                    RayFrag q;
                    F16Pack a0, a1;
                    const unsigned id = (unsigned)(wv * 64 + r);
                    a0.u = rayf[id * 4 + h * 2 + 0]; a1.u = rayf[id * 4 + h * 2 + 1];
                    q.q0 = a0.v; q.q1 = a1.v;
                    f32x16 hb, nn;
                    mf_products(gf, q, hb, nn);
                    const unsigned mine = mf_signs_behind(hb, nn, tmv0, mr0);
                    full0 = (mine << 16) | mf_from_upper_half(mine);          // lane r < 32: the word of ray r
                }
                {
                    RayFrag q;
                    F16Pack a0, a1;
                    const unsigned id = (unsigned)(wv * 64 + 32 + r);
                    a0.u = rayf[id * 4 + h * 2 + 0]; a1.u = rayf[id * 4 + h * 2 + 1];
                    q.q0 = a0.v; q.q1 = a1.v;
                    f32x16 hb, nn;
                    mf_products(gf, q, hb, nn);
                    const unsigned mine = mf_signs_behind(hb, nn, tmv1, mr1);
                    full1 = (mine << 16) | mf_from_upper_half(mine);          // lane r < 32: the word of ray 32 + r
                }
                const unsigned up = mf_from_lower_half(full1);                // lane 32 + r: the word of ray 32 + r
                const unsigned gm = own ? (h == 0 ? full0 : up) : 0u;
                // group k's rays of this wave as a 64-bit ballot; its lower word goes to lane k, its upper word to lane k + 32
#if MFB_TRANSPOSE
                // (group k is bit 31 - k of a ray's word)
                unsigned bw = mf_transpose32(__builtin_bitreverse32(gm), lane);
#else
                unsigned bw = 0;
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    const unsigned long long b = __ballot((gm & (0x80000000u >> k)) != 0u);
                    // (wait states: see the comment at the narrow form below)
                    asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                                 : "+v"(bw) : "s"((unsigned)b), "s"((unsigned)(b >> 32)), "i"(k), "i"(k + 32));
                }
#endif
                // lanes k and k + 32 share one reservation in group k's list: the lower lane asks for both
                const unsigned cnt = (unsigned)__popc(bw);
                const unsigned cnt_hi = mf_from_upper_half(cnt);
                unsigned base = 0u;
                if (h == 0 && cnt + cnt_hi != 0u) base = atomicAdd(&gcount[lane], cnt + cnt_hi);
                const unsigned base_lo = mf_from_lower_half(base), cnt_lo = mf_from_lower_half(cnt);
                unsigned place = h == 0 ? base : base_lo + cnt_lo;
                unsigned short* const lk = lists + r * MFB_LIST;
                while (bw != 0u && place < (unsigned)list_cap) {
                    const int rr = __builtin_ctz(bw);
                    bw &= bw - 1u;
                    lk[place++] = (unsigned short)(wv * 64 + h * 32 + rr);
                }
            }
#else
            // ---- every wave: its 32 rays' fragments into LDS, their group masks, the groups' lists
            const unsigned local = (unsigned)(wv * 32 + r);
            const unsigned pos = batch * RAYS + local;
            const bool valid = pos < count;
            const float4 none = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
#if MFB_KARG
            const WfParams& wa = karg_view();
#else
            const WfParams& wa = w_arg;
#endif
            float4 ra, rb;
            mf_load_ray(wa.rec[0], wa.rec[1], pos, valid, none, ra, rb);
            float tmv = __builtin_inff(), mr = 0.0f;          // the behind test's constants, in the table's scaled lengths
            if (valid && behind) { const float2 t = wa.tm[pos]; tmv = t.x * scale; mr = t.y * scale * scale; }
            if (!(tmv == tmv) || !(mr == mr)) { tmv = __builtin_inff(); mr = 0.0f; }
            const bool ordinary = mf_ordinary(ra, rb, scale);
            const RayFrag rf = mf_ray_frag(ra, rb, scale, h, ordinary);
            F16Pack p0, p1;
            p0.v = rf.q0; p1.v = rf.q1;
            rayf[local * 4 + h * 2 + 0] = p0.u;
            rayf[local * 4 + h * 2 + 1] = p1.u;
            const bool own = valid && h == 0 && ordinary;
            if (h == 0) {
                unsigned n0 = SKIP;                            // no ray here, or one whose list is marked as overflowed
                if (valid && !ordinary) { if (win == 0) cand_cnt[pos] = WF_OVERFLOW; }
                else if (valid) {
                    n0 = 0u;
                    if (win > 0) { n0 = cand_cnt[pos]; if (n0 > (unsigned)WF_CAND_CAP) n0 = (unsigned)WF_CAND_CAP + 1u; }
                }
                rcnt[local] = n0;
            }
            unsigned gm;
            {
                f32x16 hb, nn;
                mf_products(gfr[gn * MF_GROUP_WORDS + lane], rf, hb, nn);
                const unsigned mine = mf_signs_behind(hb, nn, tmv, mr);
                const unsigned full = (mine << 16) | mf_from_upper_half(mine);
                gm = own ? full : 0u;
            }
            // group k's rays of this wave as a ballot word, kept by lane k; one LDS atomic (all 32 groups at once)
            // reserves the wave's places in the groups' lists
            unsigned bw = 0;                                   // lane k < 32: bit r set iff ray r of the wave keeps group k
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const unsigned b = (unsigned)__ballot((gm & (0x80000000u >> k)) != 0u);     // (upper lane half: gm = 0)
#if MFB_WRITELANE
                // The word goes straight into lane k: one instruction, and no SGPR pair stays live per group (the select form
                // kept all 32 ballots in SGPRs: 62 of them spilled).  The wait states are needed -- the ballot's SGPR was
                // written by a vector compare; without them the lane received a stale word on the hardware -- and the
                // compiler inserts none inside an asm statement (5 covers every VALU-writes-SGPR case of the ISA manual).
                asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(bw) : "s"(b), "i"(k));
#else
                if (lane == k) bw = b;
#endif
            }
            // lane k enters group k's rays of this wave in the group's list (places past the list's end stay unwritten:
            // the group is then scanned for all rays of the pass)
            if (lane < 32 && bw != 0u) {
                unsigned place = atomicAdd(&gcount[lane], (unsigned)__popc(bw));
                unsigned short* const lk = lists + lane * MFB_LIST;
                while (bw != 0u && place < (unsigned)list_cap) {
                    const int rr = __builtin_ctz(bw);
                    bw &= bw - 1u;
                    lk[place++] = (unsigned short)(wv * 32 + rr);
                }
            }
#endif
            __syncthreads();                                   // the lists are complete
            // ---- items (group, 32 rays): lane g < 32 holds group g's range of item numbers.  A group whose list did
            // not hold all its rays takes the pass's rays as they come (NW items), no list.
            const unsigned cg = (lane < gn) ? gcount[lane] : 0u;
            const bool dense = cg > (unsigned)list_cap;
            const unsigned chunks = dense ? (unsigned)NW : (cg + 31u) >> 5;
            const unsigned incl = mf_prefix32(chunks);
            const unsigned start = incl - chunks;
            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 31);
            if (threadIdx.x < MFB_GROUPS) gcount2[((pass & 1u) ^ 1u) * MFB_GROUPS + threadIdx.x] = 0u;     // the next pass's counters
            tiles_done += (MFB_WIDE ? (wv < NBUILD ? 2u : 0u) : 1u) + (total > (unsigned)wv ? (total - (unsigned)wv + NW - 1u) / NW : 0u);      // group records + this wave's items
#if MFB_KARG
            const WfParams& wc = karg_view();
#else
            const WfParams& wc = w_arg;
#endif
            uint2* const cand = wc.cand;
            const size_t cstride = wc.slots;
            for (unsigned item = (unsigned)wv; item < total; item += NW) {     // items cost the same: dealt out in turn
                // the item's group: the first whose range ends past it (the ends ascend over the lanes 0..31)
                const int g = (int)__popcll(__ballot(incl <= item) & 0xffffffffull);
                const unsigned c = item - (unsigned)__builtin_amdgcn_readlane((int)start, g);
                const unsigned cgg = (unsigned)__builtin_amdgcn_readlane((int)cg, g);
                const bool all_rays = cgg > (unsigned)list_cap;
                const unsigned n_in = all_rays ? 32u : cgg - c * 32u;
                const bool have = (unsigned)r < n_in;
                unsigned id = c * 32u + (unsigned)r;
                if (!all_rays) id = have ? (unsigned)lists[g * MFB_LIST + id] : (unsigned)RAYS;
                RayFrag q;
                F16Pack a0, a1;
                a0.u = rayf[id * 4 + h * 2 + 0]; a1.u = rayf[id * 4 + h * 2 + 1];
                q.q0 = a0.v; q.q1 = a1.v;
                f32x16 hb, nn;
                mf_products(gfr[g * MF_GROUP_WORDS + lane], q, hb, nn);
                const unsigned mine = mf_signs(hb, nn);
                const unsigned full = (mine << 16) | mf_from_upper_half(mine);
                // (a ray without a list of its own -- none here, not ordinary -- has zero fragments in a dense group's item; its
                // counter starts at SKIP, far past the capacity: no read of the counter ahead of the atomic)
                if (h == 0 && have && full != 0u) {
                    const unsigned at = atomicAdd(&rcnt[id], 1u);
                    if (at < (unsigned)WF_CAND_CAP) cand[(size_t)at * cstride + (batch * RAYS + id)] = make_uint2((unsigned)(g0 + g), full);
                }
            }
            __syncthreads();                                   // every item is done
            // ---- the rays' list lengths, by the lanes that own the rays (they set the next pass's start values next)
            if (MFB_WIDE ? valid : h == 0) {
                const unsigned n1 = rcnt[local];
                if (n1 < SKIP) cand_cnt[pos] = n1 > (unsigned)WF_CAND_CAP ? WF_OVERFLOW : n1;
            }
        }
    }
    // one atomic per workgroup on the product counter
    __syncthreads();
    unsigned* const tiles_wg = gcount2;
    if (threadIdx.x == 0) *tiles_wg = 0u;
    __syncthreads();
    if (lane == 0) atomicAdd(tiles_wg, tiles_done);
    __syncthreads();
    if (threadIdx.x == 0 && *tiles_wg) atomicAdd((unsigned long long*)(w.kp.ctrl + 14), (unsigned long long)*tiles_wg);
}

// self-test: keep masks of n_rays rays (records of broad_setup in ra/rb) against n_groups groups, masks[ray * n_groups + g]
__global__ __launch_bounds__(64) void selftest_mfma_kernel(int n_rays, const float4* ra, const float4* rb, const uint4* frags, int n_groups, float scale, unsigned* masks) {
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    for (int tile = blockIdx.x; tile * 32 < n_rays; tile += gridDim.x) {
        const int ray = tile * 32 + r;
        const bool valid = ray < n_rays;
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
        const float4 a = valid ? ra[ray] : z, b = valid ? rb[ray] : z;
        const bool ordinary = mf_ordinary(a, b, scale);
        const RayFrag rf = mf_ray_frag(a, b, scale, h, ordinary);
        for (int g = 0; g < n_groups; g++) {
            f32x16 hb, nn;
            mf_products(frags[(size_t)g * MF_GROUP_WORDS + lane], rf, hb, nn);
            const unsigned mine = mf_signs(hb, nn);
            const unsigned other = mf_from_upper_half(mine);
            if (valid && h == 0) masks[(size_t)ray * n_groups + g] = ordinary ? ((mine << 16) | other) : 0xffffffffu;
        }
    }
}

__global__ void selftest_rayrec_kernel(int n, const double* o, const double* d, float cmax, int wide, float4* ra, float4* rb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double T[3] = {0.0, 0.0, 0.0};
    const RayRec r = broad_setup(dmk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dmk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), T, cmax, wide != 0);
    ra[i] = r.a; rb[i] = r.b;
}

}  // namespace
