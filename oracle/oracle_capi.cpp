// TEST INFRASTRUCTURE ONLY -- C entry points of the CPU oracle for ctypes.
// Loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
// product library never links or loads this file.
#include "rt_oracle.hpp"

using namespace rto;

extern "C" {

// policy: 0 = RefPolicy (double, libm), 1 = DevPolicy (double, portable sin/cos/acos/pow).
// rng_mode: 0 = per-thread streams (the reference's order), 1 = per-pixel streams.
// Returns 0, -1 if a loop guard tripped (scene outside the reference's defined
// behaviour), -2 on bad arguments.
int oracle_render(const double* params, int n_prims, const double* prims, int n_mats, const double* mats,
                  int n_verts, const double* verts, int policy, int rng_mode, int n_workers,
                  int row_begin, int row_end, double* rgb, int64_t* rays, int32_t* hit_ids, int32_t* pix_rays) {
    if (!params || !prims || !mats || !rgb || !rays) return -2;
    SceneD sc;
    sc.n_prims = n_prims; sc.prims = prims; sc.n_mats = n_mats; sc.mats = mats; sc.n_verts = n_verts; sc.verts = verts;
    if (policy == 0) return render<RefPolicy>(sc, params, rng_mode, n_workers, row_begin, row_end, rgb, rays, hit_ids, pix_rays);
    if (policy == 1) return render<DevPolicy>(sc, params, rng_mode, n_workers, row_begin, row_end, rgb, rays, hit_ids, pix_rays);
    return -2;
}

// Debugging aid: one pixel in per-pixel RNG order (device policy), every segment printed to stderr.
int oracle_trace_pixel(const double* params, int n_prims, const double* prims, int n_mats, const double* mats,
                       int n_verts, const double* verts, int x, int y, double* rgb) {
    SceneD sc;
    sc.n_prims = n_prims; sc.prims = prims; sc.n_mats = n_mats; sc.mats = mats; sc.n_verts = n_verts; sc.verts = verts;
    Tracer<DevPolicy> tr;
    tr.load(sc, params);
    tr.log = stderr;
    Rng g; g.s = pixel_seed((uint32_t)(y * tr.width + x), tr.frame_seed);
    long rays = 0;
    double out[3];
    tr.pixel(x, y, g, &rays, out, nullptr);
    for (int k = 0; k < 3; k++) rgb[k] = out[k];
    return tr.guard_tripped ? -1 : 0;
}

void oracle_expose(int width, int height, double exposure, int num_blurs, int n_workers, double* rgb, int32_t* img) {
    expose(width, height, exposure, num_blurs, n_workers, rgb, img);
}

uint32_t oracle_pixel_seed(uint32_t pixel_index, uint32_t frame_seed) { return pixel_seed(pixel_index, frame_seed); }

void oracle_rng_stream(uint32_t state, int n, uint32_t* out) {
    Rng g; g.s = state;
    for (int i = 0; i < n; i++) out[i] = g.next();
}

// fn: 0 sin, 1 cos, 2 acos, 3 pow(x,y)
void oracle_math(int fn, int n, const double* x, const double* y, double* out) {
    for (int i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = orm::p_sin(x[i]); break;
            case 1: out[i] = orm::p_cos(x[i]); break;
            case 2: out[i] = orm::p_acos(x[i]); break;
            default: out[i] = orm::p_pow(x[i], y[i]); break;
        }
    }
}

// Function-level known answers: for n rays (in: 7 doubles each -- o[3], d[3], RNG state) the records that
// oracle/ref_kat.cpp obtains from the reference's own functions, in the same layout (record doubles = 35 + 3*n_prims):
// every solid's intersect point, the closest hit with normal and material, the sampled reflection ray, the two
// refraction rays, and one whole path (colour, ray count, RNG state after).  RefPolicy (double, libm).
void oracle_kat(const double* params, int n_prims, const double* prims, int n_mats, const double* mats,
                int n_verts, const double* verts, int n, const double* in, double* out) {
    typedef Tracer<RefPolicy> T;
    SceneD sc;
    sc.n_prims = n_prims; sc.prims = prims; sc.n_mats = n_mats; sc.mats = mats; sc.n_verts = n_verts; sc.verts = verts;
    T tr;
    tr.load(sc, params);
    const int stride = 35 + 3 * n_prims;
    for (int r = 0; r < n; r++) {
        const double* q = in + 7 * (size_t)r;
        double* w = out + (size_t)stride * r;
        const T::V o = T::mk(q[0], q[1], q[2]), d = T::mk(q[3], q[4], q[5]);
        const uint32_t state = (uint32_t)q[6];
        for (int k = 0; k < 7; k++) *w++ = q[k];
        for (int i = 0; i < n_prims; i++) {
            const T::Prim& pr = tr.prims[i];
            T::V p0;
            if (pr.kind == KIND_SPHERE) p0 = T::sphere_intersect(pr, o, d);
            else if (pr.kind == KIND_PLANE) p0 = T::plane_intersect(pr.a, pr.n, o, d);
            else p0 = T::face_intersect(pr, o, d);
            for (int k = 0; k < 3; k++) *w++ = p0.v[k];
        }
        T::V p = T::mk(T::INF(), T::INF(), T::INF()), normal = T::mk(0, 0, 0);
        int mi = -1;
        tr.closest(o, d, &p, &normal, &mi);
        for (int k = 0; k < 3; k++) *w++ = p.v[k];
        for (int k = 0; k < 3; k++) *w++ = normal.v[k];
        // the reference starts from an all-zero Material (main.cpp:129)
        *w++ = mi >= 0 ? tr.mats[mi].gloss : 0.0;
        *w++ = mi >= 0 ? tr.mats[mi].rindex : 0.0;
        *w++ = mi >= 0 ? tr.mats[mi].sexp : 0.0;
        *w++ = mi >= 0 ? tr.mats[mi].diff[0] : 0.0;
        const bool front = T::valid(p) && T::dot(normal, d) < 0;
        if (front && !tr.mats[mi].is_light) {
            const T::Mat& m = tr.mats[mi];
            Rng g; g.s = state;
            const T::V rd = tr.reflection_dir(d, normal, m, g);
            for (int k = 0; k < 3; k++) *w++ = p.v[k];
            for (int k = 0; k < 3; k++) *w++ = rd.v[k];
            *w++ = (double)g.s;
            const T::V ed = T::refraction_dir(d, normal, m, 1.0);
            for (int k = 0; k < 3; k++) *w++ = ed.v[k];
            const T::V ld = T::refraction_dir(d, normal, m, m.rindex);
            for (int k = 0; k < 3; k++) *w++ = ld.v[k];
        } else {
            for (int k = 0; k < 13; k++) *w++ = 0.0;
        }
        if (front || !T::valid(p)) {
            Rng g; g.s = state;
            long rays = 0;
            double c[3];
            tr.trace(o, d, g, &rays, nullptr, c);
            for (int k = 0; k < 3; k++) *w++ = c[k];
            *w++ = (double)rays;
            *w++ = (double)g.s;
        } else {
            for (int k = 0; k < 5; k++) *w++ = 0.0;
        }
    }
}

}  // extern "C"
