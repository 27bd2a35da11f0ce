// Pipeline, part 1: path state in memory, init_kernel, broad_kernel.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

// ---- three-kernel pipeline (per-pixel mode) ----------------------------------------------------
// Path state lives in global memory, one slot per pixel of the batch, structure-of-arrays.  A frame is
//   init_kernel                     camera ray of every pixel, queue 0 = all slots
//   repeat: broad_kernel            FP32 scan of the LDS-resident records for every queued ray
//           shade_kernel            FP64 narrow phase + segment(); surviving rays are packed into the
//                                   next queue by ballot rank (one atomic per wave)
// until a queue is empty.  Each kernel gets its own register allocation: the hot broad_kernel stays
// small and runs at high occupancy, the branchy double-precision shade_kernel may use what it needs.
#ifndef DIAG_TWICE_ALWAYS      // diagnostic builds: run a section of shade_kernel twice
#define DIAG_TWICE_ALWAYS 0
#endif
#ifndef DIAG_TWICE_CAND
#define DIAG_TWICE_CAND 0
#endif
#ifndef WALK_DEBUG
#define WALK_DEBUG 0
#endif
// shade_kernel workgroup: 256 threads.  Larger groups mean fewer queue atomics but longer waits at the compaction
// barrier (with 16-byte table reads in place: 512 threads 4.65 ms, 256 threads 4.40 ms, 128 threads 4.5 ms per C3 frame)
#ifndef SHADE_BLOCK
#define SHADE_BLOCK 256
#endif
#ifndef SHADE_WAVES
#define SHADE_WAVES 4
#endif
constexpr int BROAD_GROUP = 32;        // records per candidate-mask word; LDS windows are multiples of it
constexpr int BROAD_INNER = 8;         // records loaded from LDS per inner step of broad_kernel
#ifndef WF_CAND_CAP_N
#define WF_CAND_CAP_N 16
#endif
constexpr int WF_CAND_CAP = WF_CAND_CAP_N;        // non-zero mask words kept per ray; beyond that the ray gets a full exact scan
constexpr unsigned WF_OVERFLOW = 0xffffffffu;
// per-slot state kept between rounds: 4 ints + 12 doubles (+ 3 doubles of radiance sum, touched at path ends only).
// Not stored: pixel coordinates (a function of the slot), the current refraction index (1, or the index of the
// previous vertex's material when the path is inside a solid: a flag).
enum { SI_RNG = 0, SI_SAMPLE, SI_PACK, SI_PVMAT, SI_COUNT };     // PACK = remaining (signed 8) | nverts << 8 | flags << 16
// the 12 doubles as 6 arrays of pairs (16-byte accesses): (o.x o.y) (o.z d.x) (d.y d.z) (pvn.x pvn.y) (pvn.z pvr.x) (pvr.y pvr.z)
enum { SD_COUNT = 12, SD_PAIRS = 6 };
enum { FL_INSIDE = 4 };

struct WfParams {
    KParams kp;
    unsigned slots;            // pixel slots of this batch
    unsigned item_base;        // first local pixel index of the batch
    int iter;                  // iteration number: queue iter&1 is read, (iter+1)&1 written
    int* si[2];                // [SI_COUNT][slots] by queue position, one copy per queue: a surviving ray's state moves
    double2* sd[2];            // [SD_PAIRS][slots]   with it to its place in the next queue, so both sides are coalesced
    double* acc;               // [3][slots] radiance sum of the slot's pixel (used when paths_per_pixel > 1)
    double* stack;             // [slots][max_verts][4]: throughput factors of the slot's path, 32 bytes per vertex
    unsigned* queue[2];        // slot ids of the live rays, by queue position
    unsigned* counts;          // counts[i] = live rays entering iteration i
    float4* rec[2];            // ray records by queue position
    float2* tm;                // by queue position: the constants of the behind test on group records (RayRec::c.z, c.w), rounds after the first
    uint2* cand;               // [WF_CAND_CAP][slots]: entry e of the ray at queue position pos is cand[e * slots + pos]
                               // (neighbouring lanes read neighbouring words): (word index j, keep mask of records 32j..32j+31,
                               // bit 31-k = record 32j+k), only the non-zero words, ascending j
    unsigned* cand_cnt;        // [2][slots] (one per queue: a round's shade_kernel reads one and clears the other) entries used; above WF_CAND_CAP (WF_OVERFLOW as a rule): more non-zero words than fit
};

// A kernel's parameters seen through the kernel-argument segment (WfParams is the kernels' first argument: offset 0),
// behind a pointer the compiler cannot see through: what a section of a kernel needs of them is then loaded (scalar
// loads) where the section begins, instead of sitting in SGPRs -- and, for want of those, in spilled lanes -- for a whole
// trip of the kernel's loop.
typedef const WfParams __attribute__((address_space(4)))* WfKargPtr;
DEV const WfParams& karg_view() {
    WfKargPtr p = (WfKargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const WfParams*)p;
}

// the four state words of the ray at queue position pos
struct StateWords { int rng, sample, pack, pv_mat; };
DEV StateWords load_state_words(const WfParams& w, int q, unsigned pos) {
    const size_t n = w.slots;
    const int* si = w.si[q] + pos;
    StateWords v;
    v.rng = si[SI_RNG * n]; v.sample = si[SI_SAMPLE * n]; v.pack = si[SI_PACK * n]; v.pv_mat = si[SI_PVMAT * n];
    return v;
}
// the ray itself (o, d) and the state words were read before the closest-hit search and are passed in
DEV void load_state(const WfParams& w, int q, unsigned pos, unsigned slot, d3 ro, d3 rd, const StateWords& sw, PathState& s) {
    const size_t n = w.slots;
    s.rng = (unsigned)sw.rng;
    s.sample = sw.sample;
    const int pk = sw.pack;
    s.pv_mat = sw.pv_mat;
    s.remaining = (int)(signed char)(pk & 0xff);
    s.nverts = (pk >> 8) & 0xff;
    s.flags = (pk >> 16) & 0xff;
    map_item_to_pixel(w.item_base + slot, w.kp, s.px, s.py);
    s.out_index = w.item_base + slot;
    s.n_cur = (s.flags & FL_INSIDE) ? w.kp.mats[s.pv_mat].rindex : 1.0;
    s.acc0 = s.acc1 = s.acc2 = 0.0;
    s.acc_mem = w.acc + slot; s.acc_stride = n;
    const double2* d = w.sd[q] + pos;
    const double2 p3 = d[3 * n], p4 = d[4 * n], p5 = d[5 * n];
    s.o = ro;
    s.d = rd;
    s.pv_n = dmk(p3.x, p3.y, p4.x);
    s.pv_r = dmk(p4.y, p5.x, p5.y);
}

DEV void store_state(const WfParams& w, int q, unsigned pos, const PathState& s) {
    const size_t n = w.slots;
    int* si = w.si[q] + pos;
    si[SI_RNG * n] = (int)s.rng;
    si[SI_SAMPLE * n] = s.sample;
    // the refraction index is 1 unless the last bounce entered a solid, and then it is that solid's (= pv_mat's) index
    const int flags = (s.flags & (FL_FIRST_SEG | FL_PV_REFLECT)) | ((s.n_cur != 1.0) ? FL_INSIDE : 0);
    si[SI_PACK * n] = (s.remaining & 0xff) | ((s.nverts & 0xff) << 8) | (flags << 16);
    si[SI_PVMAT * n] = s.pv_mat;
    double2* d = w.sd[q] + pos;
    d[0] = make_double2(s.o.x, s.o.y); d[n] = make_double2(s.o.z, s.d.x); d[2 * n] = make_double2(s.d.y, s.d.z);
    d[3 * n] = make_double2(s.pv_n.x, s.pv_n.y); d[4 * n] = make_double2(s.pv_n.z, s.pv_r.x); d[5 * n] = make_double2(s.pv_r.y, s.pv_r.z);
}

DEV void emit_ray(const WfParams& w, unsigned pos, unsigned slot, const PathState& s, int q) {
    w.queue[q][pos] = slot;
    store_state(w, q, pos, s);
    const RayRec r = broad_setup(s.o, s.d, w.kp.T, w.kp.cmax, w.kp.wide != 0);
    w.rec[0][pos] = r.a; w.rec[1][pos] = r.b;
    w.tm[pos] = make_float2(r.c.z, r.c.w);
    w.cand_cnt[(size_t)q * w.slots + pos] = 0u;        // broad_kernel's slices add to it
}

// the state in which the pixel of `slot` starts its first path: RNG seeded from the pixel's frame index, camera ray
DEV void start_state(const WfParams& w, unsigned slot, PathState& s) {
    const unsigned item = w.item_base + slot;
    map_item_to_pixel(item, w.kp, s.px, s.py);
    s.rng = pixel_seed((unsigned)(s.py * w.kp.width + s.px), w.kp.frame_seed);
    s.pv_mat = 0; s.pv_n = dmk(0, 0, 0); s.pv_r = dmk(0, 0, 0);
    s.acc_mem = w.acc + slot; s.acc_stride = w.slots;
    begin_pixel(s, item, w.kp);
}

// Queue 0 = every pixel slot in order.  Only what broad_kernel reads is written: the float ray record and a zero entry
// count; the first round's shade_kernel rebuilds the start state itself (start_state).
// The first workgroup also sets the rounds' ray counts (all slots enter round 0, the later counts start at zero: the
// kernels that add to them come after this one in the stream) -- no launch of its own for that at a frame's start.
__global__ __launch_bounds__(256) void init_kernel(const WfParams w, unsigned n_counts) {
    if (blockIdx.x == 0) for (unsigned i = threadIdx.x; i < n_counts; i += blockDim.x) w.counts[i] = (i == 0) ? w.slots : 0u;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned slot = blockIdx.x * blockDim.x + threadIdx.x; slot < w.slots; slot += stride) {
        PathState s;
        start_state(w, slot, s);
        const RayRec r = broad_setup(s.o, s.d, w.kp.T, w.kp.cmax, w.kp.wide != 0);
        w.rec[0][slot] = r.a; w.rec[1][slot] = r.b;
        w.cand_cnt[slot] = 0u;
    }
}

// sign bit of (thr - s) shifted into a 32-record mask: kept iff s > thr.  A NaN s (NaN ray) comes out
// negative, i.e. kept -- the exact test then decides; padding records give s = -inf, never kept.
DEV unsigned push_bit(unsigned acc, float thr, float s) {
    return __builtin_amdgcn_alignbit(acc, __float_as_uint(thr - s), 31u);
}
DEV float broad_s2(const float4 rec, const float4 ra, const float4 rb) {
    const float hb = __builtin_fmaf(ra.z, rec.z, __builtin_fmaf(ra.y, rec.y, __builtin_fmaf(ra.x, rec.x, ra.w)));
    const float nq = __builtin_fmaf(rb.z, rec.z, __builtin_fmaf(rb.y, rec.y, __builtin_fmaf(rb.x, rec.x, rec.w)));
    return __builtin_fmaf(hb, hb, nq);
}

// The hot kernel.  Every lane carries two rays; the records are read from LDS at a wave-uniform
// address (broadcast), so one ds_read_b128 feeds 18 VALU instructions: 2 x (3 fma + 3 fma + fma + sub +
// alignbit).  There is no branch inside a group of 32 records; each group ends with one coalesced
// store of the 32-bit keep mask per ray.  Workgroups are persistent over ray batches, the LDS window is
// staged once per window.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void broad_kernel(const WfParams w) {
    extern __shared__ float4 lds[];
    const unsigned count = w.counts[w.iter];
    if (count == 0) return;
    const int nchunks = w.kp.nchunks;
    unsigned* const cand_cnt = w.cand_cnt + (size_t)(w.iter & 1) * w.slots;
    const unsigned per_block = BLOCK * 2;
    const unsigned nbatches = (count + per_block - 1) / per_block;
    // A round with fewer batches than workgroups (the late rounds of a frame, every round of a small share of a
    // frame) would leave most of the chip idle while a few workgroups scan the whole list (about 50 us for 1056
    // records).  The list is then cut into `slices` runs of mask words and workgroup b scans slice b / nbatches for
    // batch b % nbatches; the slices of a ray append to its list through an atomic on its (zeroed) entry count.
    unsigned slices = 1;
    if (nchunks == 1 && nbatches < gridDim.x) {
        const unsigned words = (unsigned)w.kp.chunks[0].count / BROAD_GROUP;
        slices = gridDim.x / nbatches;
        if (slices > words) slices = words;
    }
    if (blockIdx.x >= nbatches * slices) return;
    const float4 nan4 = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    for (int c = 0; c < nchunks; c++) {
        const ChunkDesc ch = w.kp.chunks[c];
        if (c > 0) __syncthreads();
        {
            const float4* src = w.kp.records + ch.gofs;
            for (int i = threadIdx.x; i < ch.count; i += BLOCK) lds[i] = src[i];
        }
        __syncthreads();
        const int word0 = ch.first_index / BROAD_GROUP;
        const unsigned slice = blockIdx.x / nbatches;
        int g_begin = 0, g_end = ch.count;
        if (slices > 1) {
            const unsigned words = (unsigned)ch.count / BROAD_GROUP;
            g_begin = (int)(words * slice / slices) * BROAD_GROUP;
            g_end = (int)(words * (slice + 1) / slices) * BROAD_GROUP;
        }
        for (unsigned batch = slices > 1 ? blockIdx.x % nbatches : blockIdx.x; batch < nbatches; batch += slices > 1 ? nbatches : gridDim.x) {
            const unsigned pA = batch * per_block + threadIdx.x, pB = pA + BLOCK;
            const bool vA = pA < count, vB = pB < count;
            // idle lanes carry NaN constants: no bit is ever set for them
            const float4 aA = vA ? w.rec[0][pA] : nan4, bA = vA ? w.rec[1][pA] : nan4;
            const float4 aB = vB ? w.rec[0][pB] : nan4, bB = vB ? w.rec[1][pB] : nan4;
            const float thrA = bA.w, thrB = bB.w;
            unsigned nA = 0, nB = 0;
            if (c > 0) { if (vA) nA = cand_cnt[pA]; if (vB) nB = cand_cnt[pB]; }
            uint2* outA = w.cand + pA;
            uint2* outB = w.cand + pB;
            const size_t cstride = w.slots;
            for (int g = g_begin; g < g_end; g += BROAD_GROUP) {
                unsigned mA = 0, mB = 0;
#pragma unroll 1
                for (int k0 = 0; k0 < BROAD_GROUP; k0 += BROAD_INNER) {
                    float4 r[BROAD_INNER];
#pragma unroll
                    for (int k = 0; k < BROAD_INNER; k++) r[k] = lds[g + k0 + k];
#pragma unroll
                    for (int k = 0; k < BROAD_INNER; k++) {
                        mA = push_bit(mA, thrA, broad_s2(r[k], aA, bA));
                        mB = push_bit(mB, thrB, broad_s2(r[k], aB, bB));
                    }
                }
                // append the non-zero words to the ray's list (idle lanes never store)
                const unsigned j = (unsigned)(word0 + g / BROAD_GROUP);
                if (slices > 1) {
                    // shared list: the count may pass the capacity, which marks the list as overflowed
                    if (mA != 0u && vA) { const unsigned at = atomicAdd(&cand_cnt[pA], 1u); if (at < (unsigned)WF_CAND_CAP) outA[at * cstride] = make_uint2(j, mA); }
                    if (mB != 0u && vB) { const unsigned at = atomicAdd(&cand_cnt[pB], 1u); if (at < (unsigned)WF_CAND_CAP) outB[at * cstride] = make_uint2(j, mB); }
                    continue;
                }
                if (mA != 0u && vA && nA != WF_OVERFLOW) {
                    if (nA < (unsigned)WF_CAND_CAP) { outA[nA * cstride] = make_uint2(j, mA); nA++; } else nA = WF_OVERFLOW;
                }
                if (mB != 0u && vB && nB != WF_OVERFLOW) {
                    if (nB < (unsigned)WF_CAND_CAP) { outB[nB * cstride] = make_uint2(j, mB); nB++; } else nB = WF_OVERFLOW;
                }
            }
            if (slices == 1) {
                if (vA) cand_cnt[pA] = nA;
                if (vB) cand_cnt[pB] = nB;
            }
        }
    }
}

}  // namespace
