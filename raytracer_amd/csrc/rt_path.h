// Per-path state, one segment of add_to_path, the exact narrow phase.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

#ifndef SEGMENT_EARLY_LOADS
#define SEGMENT_EARLY_LOADS 1
#endif

namespace {

// ---- per-path state and one segment of add_to_path -------------------------------------------
enum { FL_FIRST_SEG = 1, FL_PV_REFLECT = 2 };

struct PathState {
    unsigned rng;
    int px, py;
    int sample;
    int remaining;
    int nverts;
    int pv_mat;
    int flags;
    unsigned out_index;     // pixel slot in rgb / hit_ids
    double n_cur;
    double acc0, acc1, acc2;
    d3 o, d;                // current ray; o is also the previous vertex's point once nverts > 0
    d3 pv_n, pv_r;          // previous vertex: normal, mirror direction
    double* acc_mem;        // pipeline: the pixel's radiance sum lives in memory (element c at acc_mem[c*acc_stride])
    size_t acc_stride;      //           and is touched only when a path ends; null: kept in acc0..2
};

// the stack of per-vertex throughput factors: element (k, c) lives at base[(3k + c) * stride]
struct FactorStack {             // single-kernel renderer: a lane's private array
    double* base;
    size_t stride;
    DEV void get3(int k, double& a, double& b, double& c) const {
        a = base[(size_t)(3 * k) * stride]; b = base[(size_t)(3 * k + 1) * stride]; c = base[(size_t)(3 * k + 2) * stride];
    }
    DEV void set3(int k, double a, double b, double c) const {
        base[(size_t)(3 * k) * stride] = a; base[(size_t)(3 * k + 1) * stride] = b; base[(size_t)(3 * k + 2) * stride] = c;
    }
};
// pipeline: the pixel slot's entries are contiguous, 4 doubles (32 bytes) each: two 16-byte accesses per entry
struct FactorStackV {
    double* base;
    DEV void get3(int k, double& a, double& b, double& c) const {
        const double2 x = ld2(base + 4 * k), y = ld2(base + 4 * k + 2);
        a = x.x; b = x.y; c = y.x;
    }
    DEV void set3(int k, double a, double b, double c) const { st2(base + 4 * k, a, b); st2(base + 4 * k + 2, c, 0.0); }
};

// the camera ray of do_rays_i, main.cpp:255-270
DEV void start_path(PathState& s, const KParams& kp) {
    const double PI = 3.14159265358979323846;
    double xs = rng_uniform(s.rng) - 0.5;
    double ys = rng_uniform(s.rng) - 0.5;
    d3 pix = dmk((double)(s.px - kp.width / 2 + xs) / kp.scale, (double)(kp.height / 2 - s.py + ys) / kp.scale, 0.0);
    double theta = rng_uniform(s.rng) * 2.0 * PI;
    double rad = rng_uniform(s.rng) * kp.lens;
    double st, ct;
    rtm::sincosd(theta, &st, &ct);
    const d3 camera = dmk(kp.cam[0], kp.cam[1], kp.cam[2]);
    d3 org = dmk(camera.x + rad * ct, camera.y + rad * st, camera.z + 0.0);
    d3 D = dnorm(dsub(pix, camera));
    d3 C = dalong(camera, kp.focal, D);
    s.o = org;
    s.d = dnorm(dsub(C, org));
    s.remaining = kp.depth;
    s.n_cur = 1.0;
    s.nverts = 0;
    s.flags = FL_FIRST_SEG;
}

DEV void begin_pixel(PathState& s, unsigned item, const KParams& kp) {
    s.out_index = kp.stream_mode ? (unsigned)(s.py * kp.width + s.px) : item;
    s.sample = 0;
    s.acc0 = s.acc1 = s.acc2 = 0.0;
    if (s.acc_mem && kp.spp > 1) { s.acc_mem[0] = 0.0; s.acc_mem[s.acc_stride] = 0.0; s.acc_mem[2 * s.acc_stride] = 0.0; }
    start_path(s, kp);
}

// a path ended with radiance (c0,c1,c2): main.cpp:271-275.  Returns false when the lane's item is finished.
DEV bool end_path(PathState& s, double c0, double c1, double c2, unsigned item, unsigned long long item_rays, const KParams& kp) {
    if (s.acc_mem && kp.spp > 1) { s.acc0 = s.acc_mem[0]; s.acc1 = s.acc_mem[s.acc_stride]; s.acc2 = s.acc_mem[2 * s.acc_stride]; }
    s.acc0 = s.acc0 + c0; s.acc1 = s.acc1 + c1; s.acc2 = s.acc2 + c2;
    s.sample++;
    if (s.sample < kp.spp) {
        if (s.acc_mem) { s.acc_mem[0] = s.acc0; s.acc_mem[s.acc_stride] = s.acc1; s.acc_mem[2 * s.acc_stride] = s.acc2; }
        start_path(s, kp);
        return true;
    }
    double* out = kp.rgb + (size_t)s.out_index * 3;
    if (kp.spp == 1) { out[0] = s.acc0; out[1] = s.acc1; out[2] = s.acc2; }      // x / 1 = x: three divisions saved per pixel
    else { const double n = (double)kp.spp; out[0] = s.acc0 / n; out[1] = s.acc1 / n; out[2] = s.acc2 / n; }
    if (kp.stream_mode) {
        // next pixel of this reference thread: x ascending, then y += num_threads (main.cpp:250-251)
        s.px++;
        if (s.px >= kp.width) { s.px = 0; s.py += kp.n_streams; }
        if (s.py < kp.height) { begin_pixel(s, item, kp); return true; }
        kp.stream_rays[item] = item_rays;
    }
    return false;
}

// One segment of add_to_path (main.cpp:126-183) after its closest hit `h` is known: record the hit id
// of a camera ray, push the previous vertex's throughput factor (Path::trace, geometry.cpp:190-213),
// then terminate the path or sample the next ray.  Returns false when the lane's item is finished.
// POOL: the caller draws the further samples of a rejected glossy sample (lobe_retry_pool) -- *retry is then set and *lobe
// filled; the state already holds everything else of the new ray (s.pv_n is the surface normal the samples are tested
// against).  Without POOL the rejection loop runs inside.
// The words of the previous vertex's material that its throughput factor takes (Path::trace, geometry.cpp:190-213), read
// by the caller as soon as the path state names that material -- beside the other reads that follow the closest-hit
// search instead of behind them (segment() reads them itself when it is handed none).
struct PvMaterial { double2 c01, c2s0, s12; double sexp; };
DEV PvMaterial load_pv_material(const KParams& kp, int pv_mat, int flags) {
    const MatD* pm = kp.mats + pv_mat;
    const double* cs = (flags & FL_PV_REFLECT) ? pm->diff : pm->rdiff;       // (diffuse, specular) or the refraction pair
    PvMaterial v;
    v.c01 = ld2(cs); v.c2s0 = ld2(cs + 2); v.s12 = ld2(cs + 4);
    v.sexp = pm->sexp;
    return v;
}

// the hit solid's record, whole (96 bytes in five reads)
struct HitPrim { int4 ids; double2 a01, a2r, n01, n2p; };
DEV HitPrim load_hit_prim(const KParams& kp, int idx) {
    const PrimD* pi = kp.prims + idx;
    HitPrim v;
    v.ids = ldi4(&pi->kind);                           // kind, mat0, mat1, voff
    v.a01 = ld2(pi->a); v.a2r = ld2(pi->a + 2);
    v.n01 = ld2(pi->n); v.n2p = ld2(pi->n + 2);
    return v;
}

template <bool POOL, class FStack>
DEV bool segment(PathState& s, const Hit& h, const FStack& fst, unsigned item, unsigned long long item_rays,
                 const KParams& kp, LobeCtx* lobe = nullptr, bool* retry = nullptr, const PvMaterial* pvm = nullptr,
                 const HitPrim* hpre = nullptr) {
    if (s.flags & FL_FIRST_SEG) {
        if (kp.hit_ids) kp.hit_ids[(size_t)s.out_index * kp.spp + s.sample] = h.idx;
        s.flags &= ~FL_FIRST_SEG;
    }
    PROF_LANES(kp.ctrl, 2);
    if (h.idx < 0) return end_path(s, 0.0, 0.0, 0.0, item, item_rays, kp);           // main.cpp:179-182
    // The solid's record is requested whole, before its kind is known, and a material's words in one go: a read that waits
    // for another read's result costs a microsecond here, and the kernel's time is made of such chains (kind -> centre or
    // normal -> material -> exponent were four of them, now two).
#if SEGMENT_EARLY_LOADS
    const HitPrim hp = hpre ? *hpre : load_hit_prim(kp, h.idx);
    const int4 ids = hp.ids;
    const double2 a01 = hp.a01, a2r = hp.a2r, n01 = hp.n01, n2p = hp.n2p;
#else
    const PrimD* pi = kp.prims + h.idx;
    const int4 ids = ldi4(&pi->kind);                  // kind, mat0, mat1, voff
#endif
    const int kind = ids.x;
    d3 normal;
    int mi = ids.y;
    if (kind == RT_KIND_SPHERE) {
        PROF_LANES(kp.ctrl, 3);
#if !SEGMENT_EARLY_LOADS
        const double2 a01 = ld2(pi->a), a2r = ld2(pi->a + 2);
#endif
        normal = dnorm(dsub(h.p, dmk(a01.x, a01.y, a2r.x)));
    } else {
#if !SEGMENT_EARLY_LOADS
        const double2 n01 = ld2(pi->n), n2p = ld2(pi->n + 2);
#endif
        normal = dmk(n01.x, n01.y, n2p.x);
        if (kind == RT_KIND_PLANE) { PROF_LANES(kp.ctrl, 4); }
#if SEGMENT_EARLY_LOADS
        if (kind == RT_KIND_PLANE) mi = plane_material(a01.x, a2r.x, ids.y, ids.z, h.p);
#else
        if (kind == RT_KIND_PLANE) mi = plane_material(pi->a[0], pi->a[2], ids.y, ids.z, h.p);
#endif
    }
    const MatD* m = kp.mats + mi;
    const int4 mq = ldi4((const int*)m);               // is_light, gloss_thr, rindex (two words)
#if SEGMENT_EARLY_LOADS
    const double2 m_exp = ld2(&m->sexp);               // (specular_exp, 1 / specular_exp)
    // ... and the previous vertex's material beside the hit's, before the first of them is looked at (a dead end reads it
    // for nothing)
    PvMaterial pv_early;
    pv_early.c01 = pv_early.c2s0 = pv_early.s12 = make_double2(0.0, 0.0); pv_early.sexp = 1.0;
    if (!pvm && s.nverts > 0) pv_early = load_pv_material(kp, s.pv_mat, s.flags);
#endif
    const bool is_light = mq.x != 0;
    const unsigned gloss_thr = (unsigned)mq.y;
    const double rindex = __hiloint2double(mq.w, mq.z);
    const bool dead_end = (!is_light) && s.remaining == 0 && kp.has_light;             // main.cpp:137-153: always black

    // A path that ends on an emitter multiplies its factors from the last vertex back.  The last one is computed right
    // below: it stays in registers (top0..2) instead of being read back from where it has just been stored, and the one
    // before it is requested here, ahead of that computation.
    double top0 = 1.0, top1 = 1.0, top2 = 1.0, below0 = 1.0, below1 = 1.0, below2 = 1.0;
    if (is_light && s.nverts >= 2) fst.get3(s.nverts - 2, below0, below1, below2);
    if (s.nverts > 0 && !dead_end) {
        PROF_LANES(kp.ctrl, 5);
#if SEGMENT_EARLY_LOADS
        const PvMaterial pv = pvm ? *pvm : pv_early;
#else
        const PvMaterial pv = pvm ? *pvm : load_pv_material(kp, s.pv_mat, s.flags);
#endif
        d3 w = dnorm(dsub(h.p, s.o));
        double sp;
        const bool pv_reflect = (s.flags & FL_PV_REFLECT) != 0;
        const double2 c01 = pv.c01, c2s0 = pv.c2s0, s12 = pv.s12;
        if (pv_reflect) { PROF_LANES(kp.ctrl, 6); }
        if (pv_reflect) sp = rtm::powd(ddot(s.pv_r, w), pv.sexp);
        else sp = 1.0;
        double dd = ddot(s.pv_n, w);
        if (sp < 0.0) sp = 0.0;
        if (dd < 0.0) dd = 0.0;
        top0 = c01.x * dd + c2s0.y * sp; top1 = c01.y * dd + s12.x * sp; top2 = c2s0.x * dd + s12.y * sp;
        fst.set3(s.nverts - 1, top0, top1, top2);
    }

    // how the segment ends; every terminating case funnels into ONE end_path call below so that the lanes
    // of a wave that end their paths for different reasons share its instructions
    bool term = false;
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (is_light) {                                                                    // main.cpp:132-135
        const double2 e01 = ld2(m->ref), e2 = ld2(m->ref + 2);       // (ref0, ref1), (ref2, -)
        c0 = e01.x; c1 = e01.y; c2 = e2.x;
        if (s.nverts >= 1) { c0 = c0 * top0; c1 = c1 * top1; c2 = c2 * top2; }
        if (s.nverts >= 2) { c0 = c0 * below0; c1 = c1 * below1; c2 = c2 * below2; }
        for (int k = s.nverts - 3; k >= 0; k--) {
            PROF_LANES(kp.ctrl, 7);
            double f0, f1, f2;
            fst.get3(k, f0, f1, f2);
            c0 = c0 * f0; c1 = c1 * f1; c2 = c2 * f2;
        }
        term = true;
    } else if (dead_end) {
        PROF_LANES(kp.ctrl, 8);
        for (int k = 0; k < kp.light_draws; k++) rng_next(s.rng);                      // rand_point of the first light
        term = true;
    } else if (s.nverts >= kp.max_verts) {
        atomicOr(&kp.ctrl[1], ERR_STACK);
        term = true;
    } else {
        d3 fwd;
        int change = 1;
        double n2 = 1.0;
        bool reflect = false;
        // leaving (main.cpp:158-161) and entering by refraction (:168-171) call get_refraction_ray with the same arguments:
        // one call site serves both (a wave with one glass hit among its lanes pays for it once)
        const bool leaving = __builtin_fabs(rindex - s.n_cur) < 0.0001;
        if (!leaving) reflect = rindex == 0.0 || rng_next(s.rng) < 1073741823u;        // main.cpp:164-167: no draw for an opaque solid
        if (reflect) {
            PROF_LANES(kp.ctrl, 9);
#if SEGMENT_EARLY_LOADS
            fwd = reflection_dir<POOL>(s.d, normal, gloss_thr, m_exp.y, s.rng, &kp.ctrl[1], item, lobe, retry);
#else
            fwd = reflection_dir<POOL>(s.d, normal, gloss_thr, m->inv_sexp, s.rng, &kp.ctrl[1], item, lobe, retry);
#endif
        } else {
            PROF_LANES(kp.ctrl, 10);
            fwd = refraction_dir(s.d, normal, rindex, s.n_cur);
            if (leaving) change = 0; else n2 = rindex;
        }
        s.pv_n = normal;
        s.pv_r = dneg(reflect_dir(s.d, normal));
        s.pv_mat = mi;
        s.flags = reflect ? (s.flags | FL_PV_REFLECT) : (s.flags & ~FL_PV_REFLECT);
        s.nverts++;
        s.o = h.p; s.d = fwd; s.remaining -= change; s.n_cur = n2;
    }
    if (term) { PROF_LANES(kp.ctrl, 11); }
    if (term) return end_path(s, c0, c1, c2, item, item_rays, kp);
    return true;
}

// narrow phase on one candidate: the reference's double-precision test.
// The solid's first 32 bytes (point a, radius) have been read: a number in the radius word means a sphere, with no
// further read (the table holds a NaN radius for planes and faces; a sphere whose own radius is NaN takes the long
// way round and is still tested as a sphere).
DEV d3 narrow_point_from(int idx, d3 a, double radius, d3 o, d3 d, const PrimD* prims, const FaceV* verts) {
    if (radius == radius) return sphere_point(a, radius, o, d);
    // the rest of the record in one go (kind and vertex range, vertex count, normal): reads that wait for one another are
    // what a plane or a face costs.  (a, radius) ARE the record's first 32 bytes: a sphere that gets here has a NaN radius.
    const PrimD* pr = prims + idx;
    const int4 ids = ldi4(&pr->kind);                  // kind, mat0, mat1, voff
    const int vcnt = pr->vcnt;
    const double2 n01 = ld2(pr->n), n2p = ld2(pr->n + 2);
    if (ids.x == RT_KIND_SPHERE) return sphere_point(a, radius, o, d);
    const d3 n = dmk(n01.x, n01.y, n2p.x);
    if (ids.x == RT_KIND_PLANE) return plane_point(a, n, o, d);
    return face_point(a, n, vcnt, verts + ids.w, o, d);
}
DEV d3 narrow_point(int idx, d3 o, d3 d, const PrimD* prims, const FaceV* verts) {
    const PrimD* pr = prims + idx;
    const double2 a01 = ld2(pr->a), a2r = ld2(pr->a + 2);
    return narrow_point_from(idx, dmk(a01.x, a01.y, a2r.x), a2r.y, o, d, prims, verts);
}
DEV void narrow_one(int idx, d3 o, d3 d, const KParams& kp, Hit& h) { consider(narrow_point(idx, o, d, kp.prims, kp.verts), idx, o, h); }

// A candidate named by its place in a list's record order.  Lists of the matrix-core scan name places of that scan's
// own order: the solid's first 32 bytes come from a compact copy in that order (kp.mf_cand) and its file index from
// kp.mf_perm, two reads that do not wait for one another; places that hold no solid read as a NaN radius and an index
// past the list (PLACES).  Other lists name file indices (places past the list are padding and are not read).
template <bool PLACES>
DEV void narrow_place(int place, d3 o, d3 d, const KParams& kp, Hit& h) {
    if (!PLACES) { if (place < kp.n_prims) narrow_one(place, o, d, kp, h); return; }
    const double* c = kp.mf_cand + 4 * (size_t)place;
    const int idx = (int)kp.mf_perm[place];
    const double2 a01 = ld2(c), a2r = ld2(c + 2);
    if (!(a2r.y == a2r.y) && idx >= kp.n_prims) return;
    consider(narrow_point_from(idx, dmk(a01.x, a01.y, a2r.x), a2r.y, o, d, kp.prims, kp.verts), idx, o, h);
}

// Exact test of every "always" solid, read from `tab` (LDS or global) at wave-uniform addresses.  The table holds first
// n_box "box" entries of ALW_BOX doubles, then the others: ALW_HEAD doubles (kind, file index, vertex count, -, point[3],
// radius, normal[3], -) and 8 doubles per face vertex (FaceV).
//
// A box entry is a plane or face whose unit normal is exactly +-(unit vector k) and, for a face, whose every edge runs
// along one coordinate axis (box_code on the host): a wall of an axis-aligned room.  For those the reference's
// operation sequence collapses, bit for bit, as long as origin, direction and hit point are finite and below 1e300:
//   * Vector::dot with a vector that has one component +-1 and two zeros returns the other vector's component (or its
//     negation): every other product is a zero and x + (+-0) = x.  So (P - o).n = +-(P_k - o_k) and n.d = +-d_k; a
//     zero result may differ in sign, which changes nothing (t = +-0 fails t >= 0.001, a zero divisor gives |t| = inf or
//     NaN, both rejected).
//   * Face::intersect's test of vertex i, (e_i.(v_i - p)) * to_next_i < 0 with e_i = (0, L, 0) say, is
//     fl(fl(L * fl(v_i.y - p.y)) * to_next_i) < 0.  The sign of a floating-point difference is exact, and with |L| >= 1
//     and |to_next_i| >= 1 (checked on the host) neither product can round to zero unless the difference is zero, so
//     the test is: sign(L * to_next_i) * (v_i.y - p.y) < 0, i.e. p.y beyond v_i.y on one side.  All vertices together:
//     lo <= p <= hi per coordinate (+-inf where nothing constrains).
// The division, the point o + t d and the squared distance are the reference's own operations.  The test is written
// without a branch (every lane divides; the outcome is a predicate) and two entries are loaded and evaluated per step.
// Entry: code (+-(k+1): normal = sign * unit vector k), file index, P_k, -, lo[3], -, hi[3], -.
constexpr int ALW_HEAD = 12, ALW_BOX = 12;

struct BoxOut { bool valid, odd; double q; d3 p; };

DEV BoxOut box_test(const double* r, int code, bool ray_ok, d3 o, d3 d) {
    const double inf = __builtin_inf(), big = 1e300;
    // the whole entry up front, and every comparison evaluated (& not &&): no branch, no load behind a comparison
    const double2 e2 = ld2(r + 2), l01 = ld2(r + 4), l2 = ld2(r + 6), h01 = ld2(r + 8), h2 = ld2(r + 10);
    const int k = (code < 0 ? -code : code) - 1;
    const double ok = k == 0 ? o.x : (k == 1 ? o.y : o.z);
    const double dk = k == 0 ? d.x : (k == 1 ? d.y : d.z);
    const double v = e2.x - ok;
    const double num = code < 0 ? -v : v, den = code < 0 ? -dk : dk;
    const double t = num / den;
    const bool t_ok = (t >= RT_EPS) & (t < inf);                 // false for NaN and -inf
    BoxOut b;
    b.p = dalong(o, t, d);
    const bool small = (__builtin_fabs(b.p.x) < big) & (__builtin_fabs(b.p.y) < big) & (__builtin_fabs(b.p.z) < big);
    const bool inside = (b.p.x >= l01.x) & (b.p.x <= h01.x) & (b.p.y >= l01.y) & (b.p.y <= h01.y) & (b.p.z >= l2.x) & (b.p.z <= h2.x);
    b.valid = ray_ok & t_ok & small & inside;
    b.odd = ray_ok & t_ok & !small;                              // huge or overflowed point: the reference's own sequence decides
    const d3 w = dsub(b.p, o);
    b.q = ddot(w, w);
    return b;
}

// consider_q under a predicate, the common outcomes without a branch
DEV void consider_if(bool valid, double q, d3 p, int idx, Hit& h) {
    const double band = 8.8817841970012523e-16;      // 2^-50, see consider_q
    const bool nearer = q < h.q * (1.0 - band);
    const bool tie = valid & !nearer & !(q > h.q * (1.0 + band));
    bool take = valid & nearer;
    if (tie) {
        const double len = __builtin_sqrt(q), hl = __builtin_sqrt(h.q);
        take = hl > len || (hl == len && idx < h.idx);
    }
    h.q = take ? q : h.q; h.idx = take ? idx : h.idx;
    h.p.x = take ? p.x : h.p.x; h.p.y = take ? p.y : h.p.y; h.p.z = take ? p.z : h.p.z;
}

DEV d3 face_verts_tab(const double* r, int nv, d3 p0) {
    bool miss = false;
    for (int i = 0; i < nv; i++) {
        const double* v = r + ALW_HEAD + 8 * i;
        double to_p = ddot(dmk(v[3], v[4], v[5]), dsub(dmk(v[0], v[1], v[2]), p0));
        if (to_p * v[6] < 0.0) miss = true;
    }
    if (miss) p0 = dmk(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
    return p0;
}

// Returns true for a ray the box entries' test is not proven for.  With REDO its box entries are then tested here through
// the general test; without it the caller has to test them (or every solid) itself.
template <bool REDO>
DEV bool narrow_always(const double* tab, int n_always, int n_box, d3 o, d3 d, const KParams& kp, Hit& h) {
    const double big = 1e300;
    // every component finite and far from overflow, so that no difference with a table coordinate (all below 1e300 too)
    // overflows and every "x * 0" of the reference's dot products is a zero; |x| < big is false for NaN as well
    const bool ray_ok = (__builtin_fabs(o.x) < big) & (__builtin_fabs(o.y) < big) & (__builtin_fabs(o.z) < big) &
                        (__builtin_fabs(d.x) < big) & (__builtin_fabs(d.y) < big) & (__builtin_fabs(d.z) < big);
    bool odd = !ray_ok & (n_box > 0);
    int a = 0;
    for (; a + 1 < n_box; a += 2) {
        const double* r0 = tab + ALW_BOX * a;
        const double* r1 = r0 + ALW_BOX;
        const int c0 = __builtin_amdgcn_readfirstlane((int)r0[0]), i0 = __builtin_amdgcn_readfirstlane((int)r0[1]);
        const int c1 = __builtin_amdgcn_readfirstlane((int)r1[0]), i1 = __builtin_amdgcn_readfirstlane((int)r1[1]);
        const BoxOut b0 = box_test(r0, c0, ray_ok, o, d);
        const BoxOut b1 = box_test(r1, c1, ray_ok, o, d);
        consider_if(b0.valid, b0.q, b0.p, i0, h);
        consider_if(b1.valid, b1.q, b1.p, i1, h);
        odd = odd | b0.odd | b1.odd;
    }
    if (a < n_box) {
        const double* r0 = tab + ALW_BOX * a;
        const int c0 = __builtin_amdgcn_readfirstlane((int)r0[0]), i0 = __builtin_amdgcn_readfirstlane((int)r0[1]);
        const BoxOut b0 = box_test(r0, c0, ray_ok, o, d);
        consider_if(b0.valid, b0.q, b0.p, i0, h);
        odd = odd | b0.odd;
    }
    if (REDO && odd) {
        // all box entries through the general test; re-considering a solid that was already accepted changes nothing (the
        // acceptance is a strict order)
        for (int b = 0; b < n_box; b++) {
            const int idx = __builtin_amdgcn_readfirstlane((int)tab[ALW_BOX * b + 1]);
            narrow_one(idx, o, d, kp, h);
        }
    }
    int off = ALW_BOX * n_box;
    for (a = n_box; a < n_always; a++) {
        const double* r = tab + off;
        const int kind = __builtin_amdgcn_readfirstlane((int)r[0]);
        const int idx = __builtin_amdgcn_readfirstlane((int)r[1]);
        const int nv = __builtin_amdgcn_readfirstlane((int)r[2]);
        d3 p0;
        if (kind == RT_KIND_SPHERE) {
            p0 = sphere_point(dmk(r[4], r[5], r[6]), r[7], o, d);
        } else {
            p0 = plane_point(dmk(r[4], r[5], r[6]), dmk(r[8], r[9], r[10]), o, d);
            if (kind == RT_KIND_FACE && !dnan(p0.x)) p0 = face_verts_tab(r, nv, p0);
        }
        consider(p0, idx, o, h);
        off += ALW_HEAD + 8 * nv;
    }
    return odd;
}

// n / d for n below 2^31 through the multiplier and shift that the host worked out for d (and checked): the divisors here are
// the same for every pixel of a render, and a quotient by a run-time divisor is some twenty instructions, several of them
// at a quarter of the rate
DEV unsigned div_by(unsigned n, unsigned d, unsigned mul, unsigned sh) { return mul ? __umulhi(n, mul) >> sh : n / d; }
DEV void map_item_to_pixel(unsigned item, const KParams& kp, int& px, int& py) {
    const int lr = (int)div_by(item, (unsigned)kp.width, kp.div_w_mul, kp.div_w_sh);
    px = (int)(item - (unsigned)lr * (unsigned)kp.width);
    const int tile = (int)div_by((unsigned)lr, (unsigned)kp.tile_rows, kp.div_t_mul, kp.div_t_sh);
    py = (tile * kp.world + kp.rank) * kp.tile_rows + (lr - tile * kp.tile_rows);
}

enum { ST_NEED = 0, ST_ALIVE = 1, ST_DEAD = 2 };

}  // namespace
