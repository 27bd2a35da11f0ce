// Pipeline, part 2: closest hit among the candidates and one path segment per queued ray (shade_kernel).
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

#ifndef SHADE_EARLY_HP
#define SHADE_EARLY_HP 1
#endif
#ifndef SHADE_EARLY_PVM
#define SHADE_EARLY_PVM 0
#endif
#ifndef SHADE_REC_AHEAD
#define SHADE_REC_AHEAD 1
#endif

namespace {

// Closest hit of the ray queued at `pos` (get_intersection, main.cpp:34-43): the always-tested solids (table `alw`, LDS)
// and the ray's kept records through the reference's exact test.  Called by whole waves.
//
// A ray that keeps very many records (a NaN ray keeps every record, a ray with a distant origin has wide error bounds;
// list overflow is the extreme case) would hold one lane busy with hundreds of exact tests while its wave, its workgroup
// (barrier) and in the end the kernel wait: 1000 tests are 0.25 ms, more than a whole round of a small frame.  Such a
// ray is tested against EVERY solid by its whole wave instead -- lane l takes solids l, l+64, ... -- and the 64 partial
// results are merged with get_intersection's order-independent acceptance (consider_q).
constexpr unsigned HEAVY_WORDS = 6, HEAVY_BITS = 48;   // shared out: lists of more than 6 mask words with more than 48 records

template <bool PLACES>
DEV Hit find_hit(const WfParams& w, unsigned pos, bool live, d3 o, d3 d, const double* alw, const double* cand_lds, const unsigned* perm_lds, bool in_lds,
                 unsigned n_list, unsigned long long& ncand_total) {
    const int lane = threadIdx.x & 63;
    Hit h;
    h.q = __builtin_inf(); h.idx = -1;
    h.p = dmk(__builtin_inf(), __builtin_inf(), __builtin_inf());
    const unsigned n = live ? n_list : 0u;          // the list's length, read by the caller beside (or ahead of) the ray
    const uint2* cl = w.cand + pos;
    const size_t cstride = w.slots;
    bool heavy = n > (unsigned)WF_CAND_CAP;     // WF_OVERFLOW, or a shared list's count that passed the capacity
    if (!heavy && n > HEAVY_WORDS) {
        unsigned c = 0;
        for (unsigned e = 0; e < n; e++) c += (unsigned)__popc(cl[e * cstride].y);
        heavy = c > HEAVY_BITS;
    }
    if (live && !heavy) {
        // the list's entries are requested two ahead: the first two before the always-tested solids, whose tests then
        // cover the loads' latency (each entry would otherwise cost a trip to memory in the middle of the scan):
        // C3 frame -0.2 ms
        uint2 cur = make_uint2(0u, 0u), nxt = make_uint2(0u, 0u);
        if (n > 0u) cur = cl[0];
        if (n > 1u) nxt = cl[cstride];
        // a ray the box entries' test is not proven for joins the heavy rays: every solid through the general test
        PROF_LANES(w.kp.ctrl, 1);
        heavy = narrow_always<false>(alw, w.kp.n_always, w.kp.n_box, o, d, w.kp, h);
#if DIAG_TWICE_ALWAYS      // diagnostic build: the section's cost is the frame time it adds (re-considering a hit changes nothing)
        __builtin_amdgcn_sched_barrier(0);
        narrow_always<false>(alw + (w.kp.n_always > 1000 ? 1 : 0), w.kp.n_always, w.kp.n_box, o, d, w.kp, h);
#endif
        ncand_total += (unsigned)w.kp.n_always;
#if DIAG_TWICE_CAND
        for (int rep = 0; rep < 2; rep++) {
        if (rep) { if (n > 0u) cur = cl[0]; if (n > 1u) nxt = cl[cstride]; }
#endif
#if SHADE_REC_AHEAD
        {
            // The candidates as one stream of places, each one's record requested while the one before it is tested: a wave's
            // lanes hold a few candidates each and a dependent read of the table costs a microsecond under load, so the
            // reads, not the tests, set the length of this loop.
            unsigned e = 0, m = 0, word = 0;
            auto next_place = [&]() -> int {
                while (m == 0u) {
                    if (e >= n) return -1;
                    word = cur.x; m = cur.y;
                    cur = nxt;
                    if (e + 2u < n) nxt = cl[(e + 2u) * cstride];
                    e++;
                    ncand_total += (unsigned)__popc(m);
                }
                const int k = __clz((int)m);
                m &= ~(0x80000000u >> k);
                return (int)word * BROAD_GROUP + k;
            };
            // records in flight: the one under test and SHADE_REC_AHEAD more (one: a second costs 12 spilled registers and
            // 0.1 ms, touching its cache line a step earlier instead 0.03 ms)
            struct Rec { int place; double2 a01, a2r; int idx; };
            auto fetch = [&]() -> Rec {
                Rec r;
                r.place = next_place();
                r.a01 = make_double2(0.0, 0.0); r.a2r = make_double2(0.0, 0.0); r.idx = 0;
                if (PLACES) {
                    if (r.place >= 0) {
                        if (in_lds) { const double* c = cand_lds + 4 * r.place; r.a01 = ld2(c); r.a2r = ld2(c + 2); r.idx = (int)perm_lds[r.place]; }
                        else { const double* c = w.kp.mf_cand + 4 * (size_t)r.place; r.a01 = ld2(c); r.a2r = ld2(c + 2); r.idx = (int)w.kp.mf_perm[r.place]; }
                    }
                } else {
                    // lists of file indices: the solid's own record; a place past the list is padding (a NaN radius and its
                    // index past the list say so below, as for a place without a solid)
                    r.idx = r.place;
                    r.a2r = make_double2(0.0, __builtin_nan(""));
                    if (r.place >= 0 && r.place < w.kp.n_prims) { const PrimD* pr = w.kp.prims + r.place; r.a01 = ld2(pr->a); r.a2r = ld2(pr->a + 2); }
                    else if (r.place >= 0) r.idx = w.kp.n_prims;
                }
                return r;
            };
            Rec q[SHADE_REC_AHEAD + 1];
#pragma unroll
            for (int k = 0; k <= SHADE_REC_AHEAD; k++) q[k] = fetch();
            while (q[0].place >= 0) {
                PROF_LANES(w.kp.ctrl, 0);
                const Rec t = q[0];
#pragma unroll
                for (int k = 0; k < SHADE_REC_AHEAD; k++) q[k] = q[k + 1];
                q[SHADE_REC_AHEAD] = fetch();
                // (a place that holds no solid reads as a NaN radius and an index past the list)
                if ((t.a2r.y == t.a2r.y) || t.idx < w.kp.n_prims)
                    consider(narrow_point_from(t.idx, dmk(t.a01.x, t.a01.y, t.a2r.x), t.a2r.y, o, d, w.kp.prims, w.kp.verts), t.idx, o, h);
            }
        }
#else
        for (unsigned e = 0; e < n; e++) {
            const uint2 ent = cur;
            cur = nxt;
            if (e + 2u < n) nxt = cl[(e + 2u) * cstride];
            unsigned m = ent.y;
            ncand_total += (unsigned)__popc(m);
            PROF_LANES(w.kp.ctrl, 16);
            while (m) {
                PROF_LANES(w.kp.ctrl, 0);
                const int k = __clz((int)m);
                m &= ~(0x80000000u >> k);
                // a ray with a NaN direction makes every s NaN and thr - NaN negative, padding records
                // included: places past the list are never dereferenced
                narrow_place<PLACES>((int)ent.x * BROAD_GROUP + k, o, d, w.kp, h);
            }
        }
#endif
#if DIAG_TWICE_CAND
        }
#endif
    }
    unsigned long long hm = __ballot(heavy);
    while (hm) {
        const int src = __builtin_ctzll(hm);
        hm &= hm - 1ull;
        const d3 ob = dmk(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
        const d3 db = dmk(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
        Hit hl;
        hl.q = __builtin_inf(); hl.idx = -1;
        hl.p = dmk(__builtin_inf(), __builtin_inf(), __builtin_inf());
        PROF_LANES(w.kp.ctrl, 13);
        for (int i = lane; i < w.kp.n_prims; i += 64) narrow_one(i, ob, db, w.kp, hl);
        for (int off = 32; off > 0; off >>= 1) {
            const double oq = __shfl_xor(hl.q, off, 64);
            const int oi = __shfl_xor(hl.idx, off, 64);
            const d3 op = dmk(__shfl_xor(hl.p.x, off, 64), __shfl_xor(hl.p.y, off, 64), __shfl_xor(hl.p.z, off, 64));
            if (oi >= 0) consider_q(oq, op, oi, hl);
        }
        if (lane == src) {
            h = hl;
            ncand_total += (unsigned)w.kp.n_prims;
            atomicAdd(&w.kp.ctrl[6], 1u);
        }
    }
    return h;
}

// exact narrow phase + one segment of add_to_path per queued ray; survivors are packed into the next queue.
// FIRST: the round of the camera rays.  Queue position = pixel slot and the path state is the start state, which the
// lane rebuilds from the pixel index (the same operations as init_kernel, hence the same bits) instead of reading it:
// init_kernel then writes 36 bytes per pixel instead of 152 and this round reads none of them back.
// PLACES: the candidate lists name places of the matrix-core scan's own record order (narrow_place).
#ifndef SHADE_KARG
#define SHADE_KARG 1        // 0: the parameters as the compiler keeps them (113 to 120 spilled SGPRs, 13 to 16 spilled VGPRs; C3 frame +0.2 ms)
#endif
#if SHADE_KARG
#define SHADE_VIEW(name) const WfParams& name = karg_view()
#else
#define SHADE_VIEW(name) const WfParams& name = w_arg
#endif

template <bool FIRST, bool PLACES>
__global__ __launch_bounds__(SHADE_BLOCK, SHADE_WAVES * 256 / SHADE_BLOCK) void shade_kernel(const WfParams w_arg) {
    extern __shared__ double alw[];
    const WfParams& w = w_arg;
    const unsigned count = w.counts[w.iter];
    if (count == 0) return;
    for (int i = threadIdx.x; i < w.kp.always_doubles; i += SHADE_BLOCK) alw[i] = w.kp.always[i];
    // behind the always table: the scan table's compact records and file indices, when the host found room for them
    double* const cand_lds = alw + ((w.kp.always_doubles + 1) & ~1);
    unsigned* const perm_lds = (unsigned*)(cand_lds + 4 * w.kp.mf_places_lds);
    // (a copy per workgroup and launch: worth it from the second trip on -- the rounds of a small render have one)
    const bool in_lds = PLACES && w.kp.mf_places_lds > 0 && count > gridDim.x * blockDim.x;
    if (in_lds) {
        for (int i = threadIdx.x; i < 2 * w.kp.mf_places_lds; i += SHADE_BLOCK) st2(cand_lds + 2 * i, w.kp.mf_cand[2 * i], w.kp.mf_cand[2 * i + 1]);
        for (int i = threadIdx.x; i < w.kp.mf_places_lds; i += SHADE_BLOCK) perm_lds[i] = w.kp.mf_perm[i];
    }
    __syncthreads();
#if SHADE_PROFILE
    // wave-cycles per section (find_hit | segment | barriers | emit), summed over waves into ctrl[16..]
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tick = __builtin_amdgcn_s_memtime();
#define SHADE_TICK(k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); prof[k] += now - tick; tick = now; }
#else
#define SHADE_TICK(k)
#endif
    __shared__ unsigned wg_cnt[2][SHADE_BLOCK / 64], wg_base[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qi = w.iter & 1, qo = qi ^ 1;
    unsigned nrays = 0;
    unsigned long long ncand_total = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    // every lane of a workgroup runs the same number of trips so that the ballot below sees whole waves
    const unsigned trips = (count + stride - 1) / stride;
    for (unsigned t = 0; t < trips; t++) {
        const unsigned pos = t * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = pos < count;
        bool alive = false, retry = false;
        LobeCtx lobe;
        lobe.n = dmk(0, 0, 0); lobe.cost = lobe.sint = lobe.inv_exp = 0.0; lobe.flat = false;
        PathState s;
        s.rng = 0u; s.d = dmk(0, 0, 0); s.pv_n = dmk(0, 0, 0);      // what lobe_retry_pool reads of every lane, live or not
        unsigned slot = 0;
        SHADE_TICK(0);
        // the closest hit needs the ray only; the rest of the path state is loaded after it (fewer live registers)
        d3 ro = dmk(0, 0, 0), rd = dmk(0, 0, 1);
        unsigned rng0 = 0;
        SHADE_VIEW(w1);
        // the candidate list's length: asked for before the camera ray is worked out (first round) or with the ray's own reads
        unsigned n_list = 0u;
        if (live) n_list = w1.cand_cnt[(size_t)qi * w1.slots + pos];
        if (FIRST) {
            if (live) {
                // only the ray and the RNG state stay in registers during the closest-hit search
                PathState t0;
                slot = pos;
                start_state(w1, slot, t0);
                ro = t0.o; rd = t0.d; rng0 = t0.rng;
            }
        } else if (live) {
            const double2* sd = w1.sd[qi] + pos;
            const size_t n = w1.slots;
            const double2 p0 = sd[0], p1 = sd[n], p2 = sd[2 * n];
            ro = dmk(p0.x, p0.y, p1.x);
            rd = dmk(p1.y, p2.x, p2.y);
        }
        SHADE_VIEW(w2);
        const Hit h = find_hit<PLACES>(w2, pos, live, ro, rd, alw, cand_lds, perm_lds, in_lds, n_list, ncand_total);   // whole waves: heavy rays are shared out
        SHADE_TICK(1);
        SHADE_VIEW(w3);
        PvMaterial pvm;
        pvm.c01 = pvm.c2s0 = pvm.s12 = make_double2(0.0, 0.0); pvm.sexp = 1.0;
        HitPrim hp;
        hp.ids = make_int4(0, 0, 0, 0); hp.a01 = hp.a2r = hp.n01 = hp.n2p = make_double2(0.0, 0.0);
        if (live) {
            if (FIRST) {
                map_item_to_pixel(w3.item_base + slot, w3.kp, s.px, s.py);
                s.rng = rng0; s.sample = 0; s.remaining = w3.kp.depth; s.nverts = 0; s.pv_mat = 0; s.flags = FL_FIRST_SEG;
                s.out_index = w3.item_base + slot; s.n_cur = 1.0; s.acc0 = s.acc1 = s.acc2 = 0.0;
                s.o = ro; s.d = rd; s.pv_n = dmk(0, 0, 0); s.pv_r = dmk(0, 0, 0);
                s.acc_mem = w3.acc + slot; s.acc_stride = w3.slots;
            } else {
                PROF_LANES(w3.kp.ctrl, 15);
                // requested side by side: the hit solid's record (what the search knows suffices for that) and the path
                // state; then, side by side again, the two materials they name (the hit's inside segment())
#if SHADE_EARLY_HP
                if (h.idx >= 0) hp = load_hit_prim(w3.kp, h.idx);
#endif
                slot = w3.queue[qi][pos];
                const StateWords sw = load_state_words(w3, qi, pos);
                load_state(w3, qi, pos, slot, ro, rd, sw, s);
#if SHADE_EARLY_PVM
                if (s.nverts > 0) pvm = load_pv_material(w3.kp, s.pv_mat, s.flags);
#endif
            }
            nrays++;
            FactorStackV fst; fst.base = w3.stack + (size_t)slot * 4u * (unsigned)w3.kp.max_verts;
            alive = segment<true>(s, h, fst, w3.item_base + slot, 0ull, w3.kp, &lobe, &retry, (FIRST || !SHADE_EARLY_PVM) ? nullptr : &pvm, (FIRST || !SHADE_EARLY_HP) ? nullptr : &hp);
        }
#if SHADE_PROFILE
        { const unsigned long long rm = __ballot(retry); if (lane == 0 && rm) { atomicAdd(&w.kp.ctrl[30], 1u); atomicAdd(&w.kp.ctrl[31], (unsigned)__popcll(rm)); } }
#endif
        // glossy samples that were rejected (main.cpp:61-76): the whole wave draws the further samples of its few such rays
        SHADE_TICK(2);
        lobe_retry_pool(retry, lobe, s.pv_n, s.rng, s.d, &w.kp.ctrl[1], w.item_base + slot);
        SHADE_TICK(5);
        // pack the surviving rays into the next queue: rank within the wave's ballot, the waves' survivor counts
        // meet in LDS (double-buffered by trip parity) and the workgroup takes its range with ONE atomic --
        // an atomic per wave on the one counter costs more than a millisecond per frame
        const unsigned long long m = __ballot(alive);
        unsigned* cnt = wg_cnt[t & 1];
        if (lane == 0) cnt[wv] = (unsigned)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int k = 0; k < SHADE_BLOCK / 64; k++) tot += cnt[k];
            wg_base[t & 1] = tot ? atomicAdd(&w.counts[w.iter + 1], tot) : 0u;
        }
        __syncthreads();
        SHADE_TICK(3);
        unsigned base = wg_base[t & 1];
        for (int k = 0; k < wv; k++) base += cnt[k];
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        SHADE_VIEW(w4);
        if (alive) { PROF_LANES(w4.kp.ctrl, 12); }
        if (alive) emit_ray(w4, base + rank, slot, s, qo);
        SHADE_TICK(4);
    }
#if SHADE_PROFILE
    if (lane == 0) for (int k = 0; k < 6; k++) atomicAdd((unsigned long long*)(w.kp.ctrl + 16 + 2 * k), prof[k]);
#endif
    // ray and candidate totals: wave sums through LDS, one atomic pair per workgroup
    unsigned long long r = nrays, cn = ncand_total;
    for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off, 64); cn += __shfl_down(cn, off, 64); }
    __shared__ unsigned long long wg_tot[2][SHADE_BLOCK / 64];
    if (lane == 0) { wg_tot[0][wv] = r; wg_tot[1][wv] = cn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        r = 0; cn = 0;
        for (int k = 0; k < SHADE_BLOCK / 64; k++) { r += wg_tot[0][k]; cn += wg_tot[1][k]; }
        if (r) atomicAdd((unsigned long long*)(w.kp.ctrl + 2), r);
        if (cn) atomicAdd((unsigned long long*)(w.kp.ctrl + 4), cn);
    }
}

}  // namespace
