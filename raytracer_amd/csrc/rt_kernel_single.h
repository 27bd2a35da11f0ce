// Single-kernel renderer (reference-order validation mode, RT_MODE_SINGLE_KERNEL).
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

// ---- single-kernel renderer: persistent lanes, broad + narrow + shading in one loop ------------
// Used for the reference-order validation mode (one lane per reference thread) and selectable for
// the per-pixel mode (RT_MODE=mega); the per-pixel default is the three-kernel pipeline below.
// Closest hit of one ray per lane against the whole list (get_intersection, main.cpp:34-43) inside a persistent
// kernel: FP32 broad phase over the LDS windows, exact narrow phase on what it keeps, always-tested solids from global
// memory.  Called by every thread of the workgroup; a list of more than one window is staged window by window in
// lock-step (`lockstep`), a single window was staged once by the caller.
template <int BLOCK>
DEV void stage_window(const KParams& kp, float4* lds, int c) {
    const ChunkDesc ch = kp.chunks[c];
    const float4* src = kp.records + ch.gofs;
    for (int i = threadIdx.x; i < ch.count; i += BLOCK) lds[i] = src[i];
}

template <int BLOCK>
DEV Hit closest_hit_block(const KParams& kp, float4* lds, bool lockstep, bool alive, d3 o, d3 d, unsigned& ncand_total) {
    Hit h;
    h.q = __builtin_inf(); h.idx = -1;
    h.p = dmk(__builtin_inf(), __builtin_inf(), __builtin_inf());
    int cand[CAND_CAP];
    int ncand = 0;
    auto narrow = [&]() {
        for (int k = 0; k < ncand; k++) if (cand[k] < kp.n_prims) narrow_one(cand[k], o, d, kp, h);
        ncand_total += (unsigned)ncand;
        ncand = 0;
    };
    RayRec rr;
    if (alive) rr = broad_setup(o, d, kp.T, kp.cmax);
    for (int c = 0; c < kp.nchunks; c++) {
        if (lockstep) {
            if (c > 0) __syncthreads();
            stage_window<BLOCK>(kp, lds, c);
            __syncthreads();
        }
        if (alive) {
            const ChunkDesc ch = kp.chunks[c];
            for (int i = 0; i < ch.count; i += BROAD_UNROLL) {
                const float4 r0 = lds[i], r1 = lds[i + 1], r2 = lds[i + 2], r3 = lds[i + 3];
                float hb0, hb1, hb2, hb3, q0, q1, q2, q3;
                const float s0 = broad_s(r0, rr.a, rr.b, hb0, q0);
                const float s1 = broad_s(r1, rr.a, rr.b, hb1, q1);
                const float s2 = broad_s(r2, rr.a, rr.b, hb2, q2);
                const float s3 = broad_s(r3, rr.a, rr.b, hb3, q3);
                const float thr = rr.b.w;
                if ((s0 >= thr) | (s1 >= thr) | (s2 >= thr) | (s3 >= thr)) {
                    const int base = ch.first_index + i;
                    if (s0 >= thr && !broad_behind(hb0, q0, rr.c)) cand[ncand++] = base;
                    if (s1 >= thr && !broad_behind(hb1, q1, rr.c)) cand[ncand++] = base + 1;
                    if (s2 >= thr && !broad_behind(hb2, q2, rr.c)) cand[ncand++] = base + 2;
                    if (s3 >= thr && !broad_behind(hb3, q3, rr.c)) cand[ncand++] = base + 3;
                    if (ncand > CAND_CAP - BROAD_UNROLL) narrow();
                }
            }
        }
    }
    if (alive) {
        narrow_always<true>(kp.always, kp.n_always, kp.n_box, o, d, kp, h);
        ncand_total += (unsigned)kp.n_always;
        narrow();
    }
    return h;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void render_kernel(const KParams kp) {
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63;
    const bool lockstep = kp.nchunks > 1;

    int state = ST_NEED;
    unsigned item = 0;        // pixel mode: local pixel index; stream mode: stream number
    PathState s;
    s.rng = 1; s.px = s.py = 0; s.sample = 0; s.remaining = 0; s.nverts = 0; s.pv_mat = 0; s.flags = 0; s.out_index = 0;
    s.n_cur = 1.0; s.acc0 = s.acc1 = s.acc2 = 0.0;
    s.o = dmk(0, 0, 0); s.d = dmk(0, 0, 1); s.pv_n = dmk(0, 0, 0); s.pv_r = dmk(0, 0, 0);
    s.acc_mem = nullptr; s.acc_stride = 0;
    unsigned nrays = 0, ncand_total = 0;
    unsigned long long item_rays = 0;
    double fst_mem[RT_MAXV * 3];
    FactorStack fst; fst.base = fst_mem; fst.stride = 1;

    if (!lockstep) {
        stage_window<BLOCK>(kp, lds, 0);
        __syncthreads();
    }

    for (;;) {
        // ---- refill idle lanes from the work queue: one atomic per wave, work packed by ballot rank
        {
            const bool need = (state == ST_NEED);
            const unsigned long long m = __ballot(need);
            if (m != 0ull) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&kp.ctrl[0], (unsigned)__popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (need) {
                    item = base + rank;
                    if (item < kp.total_items) {
                        if (kp.stream_mode) {
                            s.px = 0; s.py = (int)item;
                            s.rng = 1;                                   // default-constructed / seed(0): state 1
                            item_rays = 0;
                        } else {
                            map_item_to_pixel(item, kp, s.px, s.py);
                            s.rng = pixel_seed((unsigned)(s.py * kp.width + s.px), kp.frame_seed);
                        }
                        if (s.py < kp.height) { state = ST_ALIVE; begin_pixel(s, item, kp); }
                        else { if (kp.stream_mode) kp.stream_rays[item] = 0; state = ST_NEED; }
                    } else {
                        state = ST_DEAD;
                    }
                }
            }
        }
        const bool alive = (state == ST_ALIVE);
        if (lockstep) {
            if (!__syncthreads_or(alive ? 1 : 0)) {
                if (__syncthreads_or(state == ST_NEED ? 1 : 0)) continue;   // a lane drew an empty stream: refill again
                break;
            }
        } else {
            if (__ballot(alive) == 0ull) {
                if (__ballot(state == ST_NEED) != 0ull) continue;
                break;
            }
        }

        const Hit h = closest_hit_block<BLOCK>(kp, lds, lockstep, alive, s.o, s.d, ncand_total);
        if (alive) {
            nrays++;
            item_rays++;
            if (!segment<false>(s, h, fst, item, item_rays, kp)) state = ST_NEED;
        }
    }

    // ---- counters: wave sums, one 64-bit atomic each per wave
    unsigned long long r = nrays, cn = ncand_total;
    for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off, 64); cn += __shfl_down(cn, off, 64); }
    if (lane == 0 && r) {
        atomicAdd((unsigned long long*)(kp.ctrl + 2), r);
        atomicAdd((unsigned long long*)(kp.ctrl + 4), cn);
    }
}

}  // namespace
