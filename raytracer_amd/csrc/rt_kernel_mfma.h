// Pipeline, part 1b: the broad phase on the matrix cores (broad_mfma_kernel).
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
//
// broad_kernel spends 6 of its 9 vector instructions per (ray, record) pair on two dot products, hb = a.c + od and
// nq = b.c + w.  A dot product of 32 rays with 32 records is a 32x32 matrix product, and v_mfma_f32_32x32x16_bf16 does
// one with k = 16 in 32 cycles on a pipe of its own.  bf16 keeps 8 significant bits, so every float is cut into three
// bf16 pieces x = xh + xm + xl (by truncation: the pieces are exact and their sum is x bit for bit) and the product
// a c = (ah + am + al)(ch + cm + cl) is taken as the six partial products down to 2^-16 of the largest,
//     ah ch + ah cm + am ch + ah cl + al ch + am cm        (dropped: am cl + al cm + al cl <= 3 * 2^-24 |a c|),
// each exact in the f32 accumulator.  Four MFMAs per 32 x 32 pairs (two for hb, two for thr - nq, k slots below) leave the
// vector ALU two instructions per pair -- t = thr - nq - hb^2 by one fma, its sign into the keep mask by one alignbit --
// instead of nine.  The result is used exactly like broad_kernel's: a conservative filter in front of the exact test.
// The error bounds Eh, E0 of broad_setup are widened for the dropped products and for an accumulation of up to 32 terms
// in unspecified order (BROAD_WIDE_*).
//
// k slots (lane half h = lane >> 5 supplies k = 8h .. 8h+7 of both operands; c = record centre, w = its fourth word,
// a = -d', od, nb = -2o, thr = the ray constants of broad_setup):
//   record fragment F0: h=0 [ch 1 | cm 1]    h=1 [ch 1 | cl 0]          F1: h=0 [ch wh | cm wm]   h=1 [ch wl | cl 1]
//   hb   = F0 x [ah odh | ah odm],[am odl | ah 0]   +  F0 x [al 0 | am 0],[0]
//   N    = F1 x [nbh -1 | nbh -1],[nbm -1 | nbh thh]  +  F0 x [nbl thm | nbm thl],[0]          (N = thr - nq)
// Output tile: column = ray (lane & 31), row = (reg & 3) + 8 (reg >> 2) + 4 h.  Record 16 h' + reg of a group sits in the
// row that lane half h' reads in register reg, so a lane's 16 sign bits are the keep bits of 16 consecutive records.
#pragma once

#ifndef MF_USE_SHFL
#define MF_USE_SHFL 0
#endif
#ifndef MF_PK_FMA
#define MF_PK_FMA 0
#endif
#ifndef MF_BLOCK_N
#define MF_BLOCK_N 512
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int MF_BLOCK = MF_BLOCK_N;           // threads per workgroup of broad_mfma_kernel
constexpr int MF_GROUP_BYTES = 2048;          // two fragments x 64 lanes x 16 bytes per group of 32 records
constexpr int MF_WINDOW_GROUPS = 36;          // most groups per LDS window (72 KiB: two workgroups per CU)

// the three bf16 pieces of a float as 16-bit patterns (truncation: x = h + m + l exactly; a non-finite x keeps only h)
__host__ __device__ inline void bf16_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    union { float f; unsigned u; } a, b, c;
    a.f = x;
    const unsigned hu = a.u & 0xffff0000u;
    h = hu >> 16;
    if ((a.u & 0x7f800000u) == 0x7f800000u) { m = 0; l = 0; return; }
    a.u = hu;
    b.f = x - a.f;                             // exact: at most 16 significant bits are left
    const unsigned mu = b.u & 0xffff0000u;
    m = mu >> 16;
    a.u = mu;
    c.f = b.f - a.f;                           // exact: at most 8 bits left, a bf16 value
    l = c.u >> 16;
}

union Bf16Pack { uint4 u; bf16x8 v; };
__host__ __device__ inline uint4 pack8(unsigned e0, unsigned e1, unsigned e2, unsigned e3, unsigned e4, unsigned e5, unsigned e6, unsigned e7) {
    return make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), e4 | (e5 << 16), e6 | (e7 << 16));
}
constexpr unsigned BF16_ONE = 0x3f80u, BF16_MINUS_ONE = 0xbf80u;

// the row of the output tile (= row of the record operand) in which record k of a group is placed
__host__ __device__ inline int mf_row_of_record(int k) { const int hp = k >> 4, reg = k & 15; return (reg & 3) + 8 * (reg >> 2) + 4 * hp; }

// the two fragments of lane `lane` for one group of 32 records (rec: 32 x float4 in stream order)
__host__ __device__ inline void mf_record_frags(const float* rec, int lane, uint4& f0, uint4& f1) {
    const int r = lane & 31, h = lane >> 5;
    int k = 0;
    for (int q = 0; q < 32; q++) if (mf_row_of_record(q) == r) k = q;
    const float* c = rec + 4 * k;
    unsigned ch[3], cm[3], cl[3], wh, wm, wl;
    for (int i = 0; i < 3; i++) bf16_split3(c[i], ch[i], cm[i], cl[i]);
    bf16_split3(c[3], wh, wm, wl);
    if (h == 0) {
        f0 = pack8(ch[0], ch[1], ch[2], BF16_ONE, cm[0], cm[1], cm[2], BF16_ONE);
        f1 = pack8(ch[0], ch[1], ch[2], wh, cm[0], cm[1], cm[2], wm);
    } else {
        f0 = pack8(ch[0], ch[1], ch[2], BF16_ONE, cl[0], cl[1], cl[2], 0u);
        f1 = pack8(ch[0], ch[1], ch[2], wl, cl[0], cl[1], cl[2], BF16_ONE);
    }
}

struct RayFrag { bf16x8 q0, q1, q2, q3; };

// the ray operand of the four MFMAs for this lane's half h; a ray that is not of ordinary size gets zeros (its list is
// marked as overflowed by the caller, so the exact test sees every solid)
DEV RayFrag mf_ray_frag(const float4 ra, const float4 rb, int h, bool ordinary) {
    unsigned ah[3], am[3], al[3], oh, om, ol, nh[3], nm[3], nl[3], th, tm, tl;
    const float a[3] = {ra.x, ra.y, ra.z}, nb[3] = {-rb.x, -rb.y, -rb.z};
    for (int i = 0; i < 3; i++) { bf16_split3(a[i], ah[i], am[i], al[i]); bf16_split3(nb[i], nh[i], nm[i], nl[i]); }
    bf16_split3(ra.w, oh, om, ol);
    bf16_split3(rb.w, th, tm, tl);
    Bf16Pack p0, p1, p2, p3;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    if (h == 0) {
        p0.u = pack8(ah[0], ah[1], ah[2], oh, ah[0], ah[1], ah[2], om);
        p1.u = pack8(al[0], al[1], al[2], 0u, am[0], am[1], am[2], 0u);
        p2.u = pack8(nh[0], nh[1], nh[2], BF16_MINUS_ONE, nh[0], nh[1], nh[2], BF16_MINUS_ONE);
        p3.u = pack8(nl[0], nl[1], nl[2], tm, nm[0], nm[1], nm[2], tl);
    } else {
        p0.u = pack8(am[0], am[1], am[2], ol, ah[0], ah[1], ah[2], 0u);
        p1.u = zero;
        p2.u = pack8(nm[0], nm[1], nm[2], BF16_MINUS_ONE, nh[0], nh[1], nh[2], th);
        p3.u = zero;
    }
    if (!ordinary) { p0.u = zero; p1.u = zero; p2.u = zero; p3.u = zero; }
    RayFrag f;
    f.q0 = p0.v; f.q1 = p1.v; f.q2 = p2.v; f.q3 = p3.v;
    return f;
}

// keep bits of this lane's ray against the 16 records its lane half sees: bit 15 - reg = record 16 h + reg of the group
DEV unsigned mf_keep16(const uint4 f0u, const uint4 f1u, const RayFrag& rf) {
    Bf16Pack f0, f1;
    f0.u = f0u; f1.u = f1u;
    const f32x16 zero = {0};
    f32x16 hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.v, rf.q0, zero, 0, 0, 0);
    hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.v, rf.q1, hb, 0, 0, 0);
    f32x16 n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.v, rf.q2, zero, 0, 0, 0);
    n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.v, rf.q3, n, 0, 0, 0);
    unsigned m = 0;
#if MF_PK_FMA
    typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const f32x2 x = {hb[i], hb[i + 1]}, y = {n[i], n[i + 1]};
        const f32x2 t = __builtin_elementwise_fma(-x, x, y);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(t[0]), 31u);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(t[1]), 31u);
    }
#else
#pragma unroll
    for (int i = 0; i < 16; i++) m = __builtin_amdgcn_alignbit(m, __float_as_uint(__builtin_fmaf(-hb[i], hb[i], n[i])), 31u);
#endif
    return m;
}

// a ray record of broad_setup that is not of ordinary size carries thr = -inf
DEV bool mf_ordinary(const float4 rb) { return rb.w > -__builtin_inff(); }

#ifndef MF_TILES_N
#define MF_TILES_N 2
#endif
constexpr int MF_TILES = MF_TILES_N;          // ray tiles (of 32 rays) per wave and batch

// For a lane of the lower half: the value of its partner lane (lane + 32).  v_permlane32_swap exchanges the upper half
// of its first operand with the lower half of its second, so with both operands = v the second comes back holding the
// upper half's values in the lower lanes.  (Lanes of the upper half get their own value back.)
DEV unsigned mf_from_upper_half(unsigned v) {
#if MF_USE_SHFL
    return (unsigned)__shfl_xor((int)v, 32, 64);
#else
    return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1];
#endif
}

// One workgroup's share of a window: batches of BLOCK/64 x MF_TILES x 32 rays against the groups [g_begin, g_end) staged
// in LDS.  SLICED: several workgroups scan different groups for the same rays and append through the list's counter;
// otherwise a ray's list is this lane's alone and its length stays in a register (no memory wait inside the loop).
template <int BLOCK, bool SLICED>
DEV void mf_scan_window(const WfParams& w, const uint4* mf_lds, unsigned* cand_cnt, unsigned count, unsigned nbatches,
                        unsigned batch0, unsigned batch_step, int g0, int g_begin, int g_end, bool first_window) {
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
            const float4 ra = valid ? w.rec[0][pos[t]] : none, rb = valid ? w.rec[1][pos[t]] : none;
            const bool ordinary = mf_ordinary(rb);
            rf[t] = mf_ray_frag(ra, rb, h, ordinary);
            own[t] = valid && h == 0;                 // the lower lane half appends for the ray
            n[t] = 0;
            if (own[t]) {
                if (!ordinary) { if (SLICED || first_window) cand_cnt[pos[t]] = WF_OVERFLOW; }
                else if (!first_window) { n[t] = cand_cnt[pos[t]]; if (n[t] > (unsigned)WF_CAND_CAP) n[t] = (unsigned)WF_CAND_CAP + 1u; }
            }
            if (!ordinary) own[t] = false;
            out[t] = w.cand + ((size_t)n[t] * cstride + pos[t]);      // not used by a lane that does not own a ray
        }
        for (int g = g_begin; g < g_end; g++) {
            const uint4 f0 = mf_lds[(g * 2 + 0) * 64 + lane], f1 = mf_lds[(g * 2 + 1) * 64 + lane];
            const unsigned j = (unsigned)(g0 + g);
#pragma unroll
            for (int t = 0; t < MF_TILES; t++) {
                const unsigned mine = mf_keep16(f0, f1, rf[t]);
                const unsigned other = mf_from_upper_half(mine);
                const unsigned full = (mine << 16) | other;       // meaningful in the lower lane half
                if (full != 0u && own[t]) {
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
            }
        }
        if (!SLICED) {
#pragma unroll
            for (int t = 0; t < MF_TILES; t++) if (own[t]) cand_cnt[pos[t]] = n[t] > (unsigned)WF_CAND_CAP ? WF_OVERFLOW : n[t];
        }
    }
}

// Same contract as broad_kernel: for every queued ray the non-zero 32-record keep masks, appended to its list.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void broad_mfma_kernel(const WfParams w, const uint4* __restrict__ frags, int n_groups, int win_groups) {
    extern __shared__ uint4 mf_lds[];
    const unsigned count = w.counts[w.iter];
    if (count == 0) return;
    unsigned* const cand_cnt = w.cand_cnt + (size_t)(w.iter & 1) * w.slots;
    constexpr unsigned per_block = (BLOCK / 64) * MF_TILES * 32;
    const unsigned nbatches = (count + per_block - 1) / per_block;
    const int n_windows = (n_groups + win_groups - 1) / win_groups;
    // fewer batches than workgroups: the groups of the (single) window are shared out in slices, see broad_kernel
    unsigned slices = 1;
    if (n_windows == 1 && nbatches < gridDim.x) {
        slices = gridDim.x / nbatches;
        if (slices > (unsigned)n_groups) slices = (unsigned)n_groups;
    }
    if (blockIdx.x >= nbatches * slices) return;
    for (int win = 0; win < n_windows; win++) {
        const int g0 = win * win_groups;
        const int gn = n_groups - g0 < win_groups ? n_groups - g0 : win_groups;
        if (win > 0) __syncthreads();
        for (int i = threadIdx.x; i < gn * (MF_GROUP_BYTES / 16); i += BLOCK) mf_lds[i] = frags[(size_t)g0 * (MF_GROUP_BYTES / 16) + i];
        __syncthreads();
        if (slices > 1) {
            const unsigned slice = blockIdx.x / nbatches;
            mf_scan_window<BLOCK, true>(w, mf_lds, cand_cnt, count, nbatches, blockIdx.x % nbatches, nbatches, g0,
                                        (int)((unsigned)gn * slice / slices), (int)((unsigned)gn * (slice + 1) / slices), true);
        } else {
            mf_scan_window<BLOCK, false>(w, mf_lds, cand_cnt, count, nbatches, blockIdx.x, gridDim.x, g0, 0, gn, win == 0);
        }
    }
}

// self-test: keep masks of n_rays rays (records of broad_setup in ra/rb) against n_groups groups, masks[ray * n_groups + g]
__global__ __launch_bounds__(64) void selftest_mfma_kernel(int n_rays, const float4* ra, const float4* rb, const uint4* frags, int n_groups, unsigned* masks) {
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    for (int tile = blockIdx.x; tile * 32 < n_rays; tile += gridDim.x) {
        const int ray = tile * 32 + r;
        const bool valid = ray < n_rays;
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
        const float4 a = valid ? ra[ray] : z, b = valid ? rb[ray] : z;
        const bool ordinary = mf_ordinary(b);
        const RayFrag rf = mf_ray_frag(a, b, h, ordinary);
        for (int g = 0; g < n_groups; g++) {
            const uint4 f0 = frags[(size_t)(g * 2 + 0) * 64 + lane], f1 = frags[(size_t)(g * 2 + 1) * 64 + lane];
            const unsigned mine = mf_keep16(f0, f1, rf);
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
