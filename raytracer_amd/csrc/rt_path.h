// Per-path state, one segment of add_to_path, the exact narrow phase.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

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
    const double n = (double)kp.spp;
    double* out = kp.rgb + (size_t)s.out_index * 3;
    out[0] = s.acc0 / n; out[1] = s.acc1 / n; out[2] = s.acc2 / n;
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
template <class FStack>
DEV bool segment(PathState& s, const Hit& h, const FStack& fst, unsigned item, unsigned long long item_rays,
                 const KParams& kp) {
    if (s.flags & FL_FIRST_SEG) {
        if (kp.hit_ids) kp.hit_ids[(size_t)s.out_index * kp.spp + s.sample] = h.idx;
        s.flags &= ~FL_FIRST_SEG;
    }
    if (h.idx < 0) return end_path(s, 0.0, 0.0, 0.0, item, item_rays, kp);           // main.cpp:179-182
    const PrimD* pi = kp.prims + h.idx;
    const int4 ids = ldi4(&pi->kind);                  // kind, mat0, mat1, voff
    const int kind = ids.x;
    d3 normal;
    int mi = ids.y;
    if (kind == RT_KIND_SPHERE) {
        const double2 a01 = ld2(pi->a), a2r = ld2(pi->a + 2);
        normal = dnorm(dsub(h.p, dmk(a01.x, a01.y, a2r.x)));
    } else {
        const double2 n01 = ld2(pi->n), n2p = ld2(pi->n + 2);
        normal = dmk(n01.x, n01.y, n2p.x);
        if (kind == RT_KIND_PLANE) mi = plane_material(pi->a[0], pi->a[2], ids.y, ids.z, h.p);
    }
    const MatD* m = kp.mats + mi;
    const int4 mq = ldi4((const int*)m);               // is_light, gloss_thr, rindex (two words)
    const bool is_light = mq.x != 0;
    const unsigned gloss_thr = (unsigned)mq.y;
    const double rindex = __hiloint2double(mq.w, mq.z);
    const bool dead_end = (!is_light) && s.remaining == 0 && kp.has_light;             // main.cpp:137-153: always black

    if (s.nverts > 0 && !dead_end) {
        const MatD* pm = kp.mats + s.pv_mat;
        d3 w = dnorm(dsub(h.p, s.o));
        double sp;
        const bool pv_reflect = (s.flags & FL_PV_REFLECT) != 0;
        const double* cs = pv_reflect ? pm->diff : pm->rdiff;       // (diffuse, specular) or the refraction pair
        const double2 c01 = ld2(cs), c2s0 = ld2(cs + 2), s12 = ld2(cs + 4);
        if (pv_reflect) sp = rtm::powd(ddot(s.pv_r, w), pm->sexp);
        else sp = 1.0;
        double dd = ddot(s.pv_n, w);
        if (sp < 0.0) sp = 0.0;
        if (dd < 0.0) dd = 0.0;
        fst.set3(s.nverts - 1, c01.x * dd + c2s0.y * sp, c01.y * dd + s12.x * sp, c2s0.x * dd + s12.y * sp);
    }

    // how the segment ends; every terminating case funnels into ONE end_path call below so that the lanes
    // of a wave that end their paths for different reasons share its instructions
    bool term = false;
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (is_light) {                                                                    // main.cpp:132-135
        const double2 e0 = ld2(&m->sexp), e12 = ld2(m->ref + 1);     // (sexp, ref0), (ref1, ref2)
        c0 = e0.y; c1 = e12.x; c2 = e12.y;
        for (int k = s.nverts - 1; k >= 0; k--) {
            double f0, f1, f2;
            fst.get3(k, f0, f1, f2);
            c0 = c0 * f0; c1 = c1 * f1; c2 = c2 * f2;
        }
        term = true;
    } else if (dead_end) {
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
        if (__builtin_fabs(rindex - s.n_cur) < 0.0001) {                               // leaving, main.cpp:158-161
            fwd = refraction_dir(s.d, normal, rindex, s.n_cur);
            change = 0;
        } else if (rindex == 0.0 || rng_next(s.rng) < 1073741823u) {                   // main.cpp:164-167
            fwd = reflection_dir(s.d, normal, gloss_thr, m->sexp, s.rng, &kp.ctrl[1], item);
            reflect = true;
        } else {                                                                       // main.cpp:168-171
            fwd = refraction_dir(s.d, normal, rindex, s.n_cur);
            n2 = rindex;
        }
        s.pv_n = normal;
        s.pv_r = dneg(reflect_dir(s.d, normal));
        s.pv_mat = mi;
        s.flags = reflect ? (s.flags | FL_PV_REFLECT) : (s.flags & ~FL_PV_REFLECT);
        s.nverts++;
        s.o = h.p; s.d = fwd; s.remaining -= change; s.n_cur = n2;
    }
    if (term) return end_path(s, c0, c1, c2, item, item_rays, kp);
    return true;
}

// narrow phase on one candidate: the reference's double-precision test
DEV d3 narrow_point(int idx, d3 o, d3 d, const PrimD* prims, const FaceV* verts) {
    const PrimD* pr = prims + idx;
    const double2 a01 = ld2(pr->a), a2r = ld2(pr->a + 2);
    const d3 a = dmk(a01.x, a01.y, a2r.x);
    // the table holds a NaN radius for planes and faces: a number means a sphere, with no further read (a sphere
    // whose own radius is NaN takes the long way round and is still tested as a sphere)
    if (a2r.y == a2r.y) return sphere_point(a, a2r.y, o, d);
    const int4 ids = ldi4(&pr->kind);                  // kind, mat0, mat1, voff
    if (ids.x == RT_KIND_SPHERE) return sphere_point(a, a2r.y, o, d);
    const double2 n01 = ld2(pr->n), n2p = ld2(pr->n + 2);
    const d3 n = dmk(n01.x, n01.y, n2p.x);
    if (ids.x == RT_KIND_PLANE) return plane_point(a, n, o, d);
    return face_point(a, n, pr->vcnt, verts + ids.w, o, d);
}
DEV void narrow_one(int idx, d3 o, d3 d, const KParams& kp, Hit& h) { consider(narrow_point(idx, o, d, kp.prims, kp.verts), idx, o, h); }

// exact test of every "always" solid, read from `tab` (LDS or global) at wave-uniform addresses
constexpr int ALW_HEAD = 12;
DEV void narrow_always(const double* tab, int n_always, d3 o, d3 d, Hit& h) {
    int off = 0;
    for (int a = 0; a < n_always; a++) {
        const double* r = tab + off;
        const int kind = __builtin_amdgcn_readfirstlane((int)r[0]);
        const int idx = __builtin_amdgcn_readfirstlane((int)r[1]);
        const int nv = __builtin_amdgcn_readfirstlane((int)r[2]);
        d3 p0;
        if (kind == RT_KIND_SPHERE) {
            p0 = sphere_point(dmk(r[4], r[5], r[6]), r[7], o, d);
        } else {
            p0 = plane_point(dmk(r[4], r[5], r[6]), dmk(r[8], r[9], r[10]), o, d);
            if (kind == RT_KIND_FACE && !dnan(p0.x)) {
                bool miss = false;
                for (int i = 0; i < nv; i++) {
                    const double* v = r + ALW_HEAD + 8 * i;
                    double to_p = ddot(dmk(v[3], v[4], v[5]), dsub(dmk(v[0], v[1], v[2]), p0));
                    if (to_p * v[6] < 0.0) miss = true;
                }
                if (miss) p0 = dmk(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
            }
        }
        consider(p0, idx, o, h);
        off += ALW_HEAD + 8 * nv;
    }
}

DEV void map_item_to_pixel(unsigned item, const KParams& kp, int& px, int& py) {
    const int lr = (int)(item / (unsigned)kp.width);
    px = (int)(item - (unsigned)lr * (unsigned)kp.width);
    py = ((lr / kp.tile_rows) * kp.world + kp.rank) * kp.tile_rows + lr % kp.tile_rows;
}

enum { ST_NEED = 0, ST_ALIVE = 1, ST_DEAD = 2 };

}  // namespace
