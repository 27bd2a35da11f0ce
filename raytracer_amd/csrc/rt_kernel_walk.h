// Pipeline, alternative to broad_kernel: walk_kernel (uniform grid, RT_OPT_ACCEL).
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

// ---- grid walk: the broad phase with a uniform grid over the bounded solids (option RT_OPT_ACCEL) ----------------
// get_intersection (main.cpp:34-43) tests every solid for every ray; broad_kernel does the same with a cheap test.
// walk_kernel produces candidate lists that lead to the same closest hit from far fewer tests: every lane walks its
// own ray through a uniform grid (3D-DDA, front to back), applies broad_kernel's conservative test to the solids
// listed in the cells it visits, and stops once it has left behind a sphere that the exact test is CERTAIN to accept.
// Nothing here decides a result: the exact phase still picks the closest hit among the candidates with the
// reference's own arithmetic.  What has to hold is that no candidate list lacks the solid the full scan would have
// selected:
//   * a solid is listed in every cell that its bounding box, grown by `pad`, overlaps; float rounding moves the
//     walk's cell boundaries by far less than `pad` (boundaries are recomputed from the integer cell index at every
//     step, never accumulated; rays that start further away than `far` from the grid are handed to the exact
//     full scan instead), so a point of the exact ray at parameter t lies within `pad` of the cell the walk is in
//     at t, and a solid hit there is listed in that cell;
//   * a hit is certain when the float discriminant, evaluated relative to the sphere's centre, exceeds a bound M on
//     its own error and a root exceeds the reference's 0.001 threshold by more than the bound Mt on the root's
//     error; then the exact test returns a valid point no further than tU = root + Mt.  The walk ends when the
//     current cell's exit parameter exceeds the smallest tU by more than `pad`: every solid not yet seen is hit,
//     if at all, strictly further away than the certain hit, and cannot be the closest one (nor tie with it).
// Rays with a non-finite component keep every record in broad_kernel; here they are marked as overflowed lists, for
// which the exact phase scans every solid.
struct GridD {
    float lo[3], hi[3];            // box of the grid, relative to the translation T
    float cell[3], inv_cell[3];    // cell size and its reciprocal
    int n[3];                      // cells per axis
    float pad;                     // growth of the solids' boxes at insertion; slack of the early exit
    float far;                     // largest |origin component| (relative to T) the walk accepts
    const uint4* cells;            // per cell two uint4 (32 bytes, one cache line half): the broad-phase record
                                   // (c, R^2 - |c|^2) of the cell's FIRST item as float bits, then (file index of that
                                   // item's solid, float bits of a lower bound of r^2 if it is a sphere else of -1,
                                   // items in the cell, place of the cell's further items in item_rec / item_aux).
                                   // Most cells list at most one solid: one load per step, whose address depends on
                                   // the walk's arithmetic only, so it is issued a step ahead.
    const float4* item_rec;        // further items: record
    const uint2* item_aux;         //                (file index, r^2 bound bits)
};

DEV float grid_bound(const GridD& g, int k, int i) { return __builtin_fmaf((float)i, g.cell[k], g.lo[k]); }

__global__ __launch_bounds__(256) void walk_kernel(const WfParams w, const GridD g) {
    const unsigned count = w.counts[w.iter];
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned* const cand_cnt = w.cand_cnt + (size_t)(w.iter & 1) * w.slots;
    const float inf = __builtin_inff();
    for (unsigned pos = blockIdx.x * blockDim.x + threadIdx.x; pos < count; pos += stride) {
        PROF_LANES(w.kp.ctrl, 17);
        const float4 ra = w.rec[0][pos], rb = w.rec[1][pos];
        uint2* out = w.cand + pos;
        const size_t cstride = w.slots;
        const float o[3] = {0.5f * rb.x, 0.5f * rb.y, 0.5f * rb.z};
        const float d[3] = {-ra.x * (1.0f / BROAD_SCALE), -ra.y * (1.0f / BROAD_SCALE), -ra.z * (1.0f / BROAD_SCALE)};
        const float thr = rb.w;
        // finite test by arithmetic: x - x is 0 for finite x, NaN otherwise
        const float chk = (o[0] - o[0]) + (o[1] - o[1]) + (o[2] - o[2]) + (d[0] - d[0]) + (d[1] - d[1]) + (d[2] - d[2]) + (thr - thr) + (ra.w - ra.w);
        const float omax = __builtin_fmaxf(__builtin_fabsf(o[0]), __builtin_fmaxf(__builtin_fabsf(o[1]), __builtin_fabsf(o[2])));
        if (!(chk == 0.0f) || !(omax <= g.far)) {
#if WALK_DEBUG
            atomicAdd(&w.kp.ctrl[!(chk == 0.0f) ? 8 : 9], 1u);
#endif
            cand_cnt[pos] = WF_OVERFLOW; continue;
        }

        // ---- the ray's interval inside the grid box, and the walk's state
        float t0 = 0.0f, t1 = inf;
        bool miss = false;
        float inv[3];
        int step[3], cell[3];
        for (int k = 0; k < 3; k++) {
            if (__builtin_fabsf(d[k]) < 1e-30f) {
                step[k] = 0; inv[k] = 0.0f;
                if (o[k] < g.lo[k] || o[k] > g.hi[k]) miss = true;
            } else {
                inv[k] = 1.0f / d[k];
                step[k] = d[k] > 0.0f ? 1 : -1;
                const float ta = (g.lo[k] - o[k]) * inv[k], tb = (g.hi[k] - o[k]) * inv[k];
                t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
                t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
            }
        }
        // the box is larger than every listed solid by 2 pad: a ray that the float interval test puts outside it by
        // more than its own rounding cannot hit any of them
        if (miss || t0 > t1 * 1.0001f + 4.0f * g.pad) { cand_cnt[pos] = 0u; continue; }
        float tmax[3];
        for (int k = 0; k < 3; k++) {
            const float p = __builtin_fmaf(d[k], t0, o[k]);
            int c = (int)__builtin_floorf((p - g.lo[k]) * g.inv_cell[k]);
            c = c < 0 ? 0 : (c >= g.n[k] ? g.n[k] - 1 : c);
            cell[k] = c;
            tmax[k] = step[k] == 0 ? inf : (grid_bound(g, k, c + (step[k] > 0 ? 1 : 0)) - o[k]) * inv[k];
        }

        // error scale of this ray's float tests: |o| + largest |c| + R (as in broad_setup)
        const float S = bound_sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]) * 1.0001f + w.kp.cmax;
        float tU = inf;
        unsigned n = 0, cur_j = 0xffffffffu, cur_m = 0u;

        // one listed solid: broad_kernel's test, then (spheres) the certain-hit test
        auto test_item = [&](const float4 rec, unsigned idx, float r2) {
            const float s = broad_s2(rec, ra, rb);
            if (s <= thr) return;                               // kept iff s > thr
            const unsigned j = idx / (unsigned)BROAD_GROUP, bit = 0x80000000u >> (idx % (unsigned)BROAD_GROUP);
            if (j == cur_j) cur_m |= bit;
            else {
                if (cur_m != 0u && n != WF_OVERFLOW) { if (n < (unsigned)WF_CAND_CAP) { out[n * cstride] = make_uint2(cur_j, cur_m); n++; } else n = WF_OVERFLOW; }
                cur_j = j; cur_m = bit;
            }
            if (r2 >= 0.0f) {
                const float ox = o[0] - rec.x, oy = o[1] - rec.y, oz = o[2] - rec.z;
                const float hb = __builtin_fmaf(d[2], oz, __builtin_fmaf(d[1], oy, d[0] * ox));
                const float oc2 = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
                const float disc = __builtin_fmaf(hb, hb, r2 - oc2);
                // bound of the float discriminant's error, about 3x the term-by-term count: rounding of o and c to
                // float and of their difference (S |oc|), of the products and sums (oc2), of d (hb^2 + oc2)
                const float M = (48.0f * F32_U) * __builtin_fmaf(S, bound_sqrtf(oc2), oc2) + (16.0f * F32_U) * __builtin_fmaf(hb, hb, oc2) + 1e-12f;
                if (disc > M) {
                    // (square roots within a unit in the last place: the root's error term has two units more than the eight
                    // that the correctly rounded ones were given)
                    const float sq = bound_sqrtf(disc);
                    const float Mt = 2.0f * bound_sqrtf(M) * 1.000001f + (10.0f * F32_U) * (__builtin_fabsf(hb) + sq + S) + 1e-6f;
                    const float ta = -hb - sq, tb = sq - hb;
                    const float lim = (float)RT_EPS + Mt;
                    const float th = ta > lim ? ta : (tb > lim ? tb : inf);
                    tU = __builtin_fminf(tU, th + Mt);
                }
            }
        };

        const int max_steps = g.n[0] + g.n[1] + g.n[2] + 4;
        unsigned ci = (unsigned)((cell[2] * g.n[1] + cell[1]) * g.n[0] + cell[0]);
        uint4 c0 = g.cells[2 * (size_t)ci], c1 = g.cells[2 * (size_t)ci + 1];
#if WALK_DEBUG
        int nsteps = 0; unsigned ntests = 0;
#endif
        for (int it = 0; it < max_steps; it++) {
            // the next cell: through the nearest of the three next boundaries; its record is requested before the
            // current cell's solids are tested
            PROF_LANES(w.kp.ctrl, 18);
            const int ax = (tmax[0] <= tmax[1] && tmax[0] <= tmax[2]) ? 0 : (tmax[1] <= tmax[2] ? 1 : 2);
            const float texit = ax == 0 ? tmax[0] : (ax == 1 ? tmax[1] : tmax[2]);
            const int st = ax == 0 ? step[0] : (ax == 1 ? step[1] : step[2]);
            const int c = (ax == 0 ? cell[0] : (ax == 1 ? cell[1] : cell[2])) + st;
            const int nn = ax == 0 ? g.n[0] : (ax == 1 ? g.n[1] : g.n[2]);
            const bool has_next = st != 0 && (unsigned)c < (unsigned)nn;   // st == 0: a point ray (every tmax infinite)
            uint4 n0 = c0, n1 = c1;
            if (has_next) {
                const float oo = ax == 0 ? o[0] : (ax == 1 ? o[1] : o[2]);
                const float iv = ax == 0 ? inv[0] : (ax == 1 ? inv[1] : inv[2]);
                // (the axis is a run-time value: selects, not indexing, keep the grid's parameters in scalar registers)
                const float cs = ax == 0 ? g.cell[0] : (ax == 1 ? g.cell[1] : g.cell[2]);
                const float gl = ax == 0 ? g.lo[0] : (ax == 1 ? g.lo[1] : g.lo[2]);
                const float tn = (__builtin_fmaf((float)(c + (st > 0 ? 1 : 0)), cs, gl) - oo) * iv;
                if (ax == 0) { cell[0] = c; tmax[0] = tn; } else if (ax == 1) { cell[1] = c; tmax[1] = tn; } else { cell[2] = c; tmax[2] = tn; }
                ci = (unsigned)((cell[2] * g.n[1] + cell[1]) * g.n[0] + cell[0]);
                n0 = g.cells[2 * (size_t)ci]; n1 = g.cells[2 * (size_t)ci + 1];
            }
            const unsigned cnt = c1.z;
#if WALK_DEBUG
            nsteps++; ntests += cnt;
#endif
            if (cnt != 0u) {
                PROF_LANES(w.kp.ctrl, 19);
                test_item(make_float4(__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w)), c1.x, __uint_as_float(c1.y));
                for (unsigned i = 0; i + 1u < cnt; i++) {
                    PROF_LANES(w.kp.ctrl, 20);
                    const uint2 aux = g.item_aux[c1.w + i];
                    test_item(g.item_rec[c1.w + i], aux.x, __uint_as_float(aux.y));
                }
            }
            if (!has_next || texit - g.pad > tU) break;
            c0 = n0; c1 = n1;
        }
        if (cur_m != 0u && n != WF_OVERFLOW) { if (n < (unsigned)WF_CAND_CAP) { out[n * cstride] = make_uint2(cur_j, cur_m); n++; } else n = WF_OVERFLOW; }
#if WALK_DEBUG
        if (n == WF_OVERFLOW) atomicAdd(&w.kp.ctrl[10], 1u);
        atomicAdd(&w.kp.ctrl[11], (unsigned)nsteps);
        atomicAdd(&w.kp.ctrl[12], ntests);
#endif
        cand_cnt[pos] = n;
    }
}

}  // namespace
