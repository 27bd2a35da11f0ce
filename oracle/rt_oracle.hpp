// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into the product library.
//
// A CPU restatement of the per-pixel render loop of samkottler/rayTracer,
// written from the reference's behaviour (flat scene tables, an iterative
// segment loop and an explicit vertex stack instead of the reference's virtual
// classes, recursion and linked list).  Each function cites the reference lines
// it follows; paths are relative to /root/reference/.
//
// The code is a template over an arithmetic policy:
//
//   RefPolicy  double, glibc libm, products and sums rounded separately.  With
//              rng_mode = STREAM this reproduces the compiled reference bit for
//              bit (pinned by tests/test_oracle_vs_reference.py against
//              oracle/_ref and by the committed fixtures under tests/golden/).
//   DevPolicy  RefPolicy with libm's sin, cos, acos and pow replaced by the portable
//              functions of rt_oracle_math.hpp -- nothing else differs.  This is
//              the arithmetic the GPU kernels implement (a GPU cannot reproduce
//              glibc bit for bit); the parity tests compare the kernels with this
//              instantiation bit for bit, and DevPolicy with RefPolicy / the
//              reference's golden images statistically.
//
// and over a run-time RNG mode:
//
//   STREAM     one minstd_rand0 stream per reference thread, thread t renders
//              rows t, t+T, ... (src/main.cpp:250,301,307).
//   PIXEL      one minstd_rand0 stream per pixel, seeded by pixel_seed(); the
//              only order a one-thread-per-pixel kernel can reproduce.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <cstdio>
#include <vector>

#include "rt_oracle_math.hpp"

namespace rto {

// ---------------------------------------------------------------- scene tables
// Flat tables of doubles as the loader (src/readShape.cpp:26-122) leaves them:
// normals already normalised, gloss already derived.
enum { KIND_SPHERE = 0, KIND_PLANE = 1, KIND_FACE = 2 };

// prim record (doubles): kind, solid_is_light, mat0, mat1, vert_off, vert_cnt,
//                        a[3] (sphere centre | plane point | face verts[0]),
//                        n[3] (plane/face normal), radius
enum { PR_KIND = 0, PR_LIGHT, PR_MAT0, PR_MAT1, PR_VOFF, PR_VCNT, PR_A, PR_N = 9, PR_RADIUS = 12, PR_STRIDE = 13 };
// material record (doubles), the fields of struct Material, src/geometry.hpp:18-29
enum { MT_REF = 0, MT_LIGHT = 3, MT_GLOSS, MT_DIFF, MT_SPEC = 8, MT_SEXP = 11, MT_RDIFF, MT_RSPEC = 15, MT_RSEXP = 18, MT_RINDEX, MT_STRIDE = 20 };
// params (doubles), the header object of the scene stream, src/readShape.cpp:31-42
enum { PA_WIDTH = 0, PA_HEIGHT, PA_SPP, PA_DEPTH, PA_THREADS, PA_BLURS, PA_SCALE, PA_LENS, PA_FOCAL, PA_EXPOSURE, PA_SEED,
       PA_CAM_X, PA_CAM_Y, PA_CAM_Z, PA_COUNT };

enum { RNG_STREAM = 0, RNG_PIXEL = 1 };

struct SceneD {
    int n_prims = 0, n_mats = 0, n_verts = 0;
    const double* prims = nullptr;
    const double* mats = nullptr;
    const double* verts = nullptr;
};

// (int)x as the reference's x86 build converts a double (cvttsd2si): truncation toward zero, INT_MIN when the value is
// out of int range or NaN.  Used wherever the reference casts a double that a scene can push out of range
// (src/geometry.cpp:108-109 checker index, :15-23 Color::to_int).
static inline int cvttsd2si(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000u;
    return (int)v;
}

// ------------------------------------------------------------------------- RNG
// std::default_random_engine of libstdc++ is minstd_rand0: x <- 16807 x mod
// (2^31 - 1), operator() returns the new state, min 1, max 2147483646; a default
// constructed engine and seed(0) both start from state 1 (src/main.cpp:21,301,307).
struct Rng {
    uint32_t s = 1;
    uint32_t next() { s = (uint32_t)(((uint64_t)s * 16807u) % 2147483647u); return s; }
};
static const uint32_t RNG_MAX = 2147483646u;

// Seed of the per-pixel stream: an integer hash of the pixel's index in the full
// frame, mapped into the generator's valid state range [1, 2^31 - 2].
static inline uint32_t pixel_seed(uint32_t pixel_index, uint32_t frame_seed) {
    uint32_t h = pixel_index * 0x9E3779B1u ^ frame_seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return 1u + (h % 2147483646u);
}

// -------------------------------------------------------------------- policies
struct RefPolicy {
    typedef double Real;
    static const bool is_ref = true;
    static Real madd(Real acc, Real a, Real b) { return acc + a * b; }
    static Real sin(Real x) { return std::sin(x); }
    static Real cos(Real x) { return std::cos(x); }
    static Real acos(Real x) { return std::acos(x); }
    static Real pow(Real x, Real y) { return std::pow(x, y); }
    static Real sqrt(Real x) { return std::sqrt(x); }
    // generators() < mat.gloss*generators.max(), src/main.cpp:48
    static bool glossy(uint32_t g, double gloss) { return (double)g < gloss * (double)RNG_MAX; }
};

struct DevPolicy {
    typedef double Real;
    static const bool is_ref = false;
    static Real madd(Real acc, Real a, Real b) { return acc + a * b; }
    static Real sin(Real x) { return orm::p_sin(x); }
    static Real cos(Real x) { return orm::p_cos(x); }
    static Real acos(Real x) { return orm::p_acos(x); }
    static Real pow(Real x, Real y) { return orm::p_pow(x, y); }
    static Real sqrt(Real x) { return __builtin_sqrt(x); }
    // same decision as RefPolicy (the device carries it as the integer threshold ceil(gloss*max))
    static bool glossy(uint32_t g, double gloss) { return (double)g < gloss * (double)RNG_MAX; }
};

// ---------------------------------------------------------------- the renderer
template <class P>
struct Tracer {
    typedef typename P::Real R;
    struct V { R v[3]; };

    struct Mat {           // struct Material, src/geometry.hpp:18-29, in policy precision
        R ref[3]; bool is_light; double gloss; R diff[3], spec[3], sexp, rdiff[3], rspec[3], rindex;
    };
    struct Prim {
        int kind; bool solid_light; int mat0, mat1; int nv; V a, n; R radius; std::vector<V> verts;
    };
    struct Vertex { V p, n, r; int mat; bool reflect; };

    // scene in policy precision
    std::vector<Prim> prims;
    std::vector<Mat> mats;
    int first_light = -1;
    int width = 0, height = 0, spp = 0, depth = 0;
    R scale = 0, lens = 0, focal = 0;
    R cam[3] = {0, 0, 40};        // main.cpp:16
    uint32_t frame_seed = 0;
    bool guard_tripped = false;
    FILE* log = nullptr;          // when set, trace() prints one line per segment (debugging aid)

    static R INF() { return std::numeric_limits<R>::infinity(); }
    static R NaN() { return std::numeric_limits<R>::quiet_NaN(); }

    // ---- Vector / Point algebra, src/Vector.cpp:10-56, src/geometry.cpp:55-64
    static V mk(R x, R y, R z) { V a; a.v[0] = x; a.v[1] = y; a.v[2] = z; return a; }
    static V vadd(const V& a, const V& b) { return mk(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
    // Vector::operator*(double s): s*nums[i]
    static V vscale(const V& a, R s) { return mk(s * a.v[0], s * a.v[1], s * a.v[2]); }
    // Vector::operator-: a + other*-1 (src/Vector.cpp:24-26)
    static V vsub(const V& a, const V& b) { return vadd(a, vscale(b, R(-1))); }
    // Point - Point: plain component subtraction (src/geometry.cpp:58-64)
    static V psub(const V& a, const V& b) { return mk(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
    // Vector::dot: total = 0; total += a[i]*b[i] (src/Vector.cpp:33-39)
    static R dot(const V& a, const V& b) {
        R t = R(0);
        for (int i = 0; i < 3; i++) t = P::madd(t, a.v[i], b.v[i]);
        return t;
    }
    static R length(const V& a) { return P::sqrt(dot(a, a)); }
    static V cross(const V& a, const V& b) {   // src/Vector.cpp:43-49
        return mk(a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]);
    }
    static V normalize(const V& a) {           // src/Vector.cpp:50-56
        R l = length(a);
        return mk(a.v[0] / l, a.v[1] / l, a.v[2] / l);
    }
    // Point + s*Vector (src/geometry.cpp:55-57 after src/Vector.cpp:67-69)
    static V along(const V& o, R s, const V& d) {
        return mk(P::madd(o.v[0], s, d.v[0]), P::madd(o.v[1], s, d.v[1]), P::madd(o.v[2], s, d.v[2]));
    }
    // Point::is_valid, src/geometry.cpp:51-54: no NaN and no +INF component (-INF passes)
    static bool valid(const V& p) {
        for (int i = 0; i < 3; i++) if (std::isnan(p.v[i]) || p.v[i] == INF()) return false;
        return true;
    }

    void load(const SceneD& s, const double* params) {
        width = (int)params[PA_WIDTH]; height = (int)params[PA_HEIGHT];
        spp = (int)params[PA_SPP]; depth = (int)params[PA_DEPTH];
        scale = (R)params[PA_SCALE]; lens = (R)params[PA_LENS]; focal = (R)params[PA_FOCAL];
        frame_seed = (uint32_t)params[PA_SEED];
        for (int k = 0; k < 3; k++) cam[k] = (R)params[PA_CAM_X + k];
        mats.resize(s.n_mats);
        for (int i = 0; i < s.n_mats; i++) {
            const double* m = s.mats + (size_t)i * MT_STRIDE;
            Mat& o = mats[i];
            for (int k = 0; k < 3; k++) {
                o.ref[k] = (R)m[MT_REF + k]; o.diff[k] = (R)m[MT_DIFF + k]; o.spec[k] = (R)m[MT_SPEC + k];
                o.rdiff[k] = (R)m[MT_RDIFF + k]; o.rspec[k] = (R)m[MT_RSPEC + k];
            }
            o.is_light = m[MT_LIGHT] != 0; o.gloss = m[MT_GLOSS];
            o.sexp = (R)m[MT_SEXP]; o.rindex = (R)m[MT_RINDEX];
        }
        prims.resize(s.n_prims);
        first_light = -1;
        for (int i = 0; i < s.n_prims; i++) {
            const double* p = s.prims + (size_t)i * PR_STRIDE;
            Prim& o = prims[i];
            o.kind = (int)p[PR_KIND]; o.solid_light = p[PR_LIGHT] != 0;
            o.mat0 = (int)p[PR_MAT0]; o.mat1 = (int)p[PR_MAT1];
            o.nv = (int)p[PR_VCNT];
            o.a = mk((R)p[PR_A], (R)p[PR_A + 1], (R)p[PR_A + 2]);
            o.n = mk((R)p[PR_N], (R)p[PR_N + 1], (R)p[PR_N + 2]);
            o.radius = (R)p[PR_RADIUS];
            o.verts.resize(o.nv);
            int off = (int)p[PR_VOFF];
            for (int k = 0; k < o.nv; k++) {
                const double* v = s.verts + (size_t)(off + k) * 3;
                o.verts[k] = mk((R)v[0], (R)v[1], (R)v[2]);
            }
            if (o.solid_light && first_light < 0) first_light = i;
        }
    }

    // ---- primitives
    // Sphere::intersect, src/geometry.cpp:129-138
    static V sphere_intersect(const Prim& s, const V& o, const V& d) {
        V oc = psub(o, s.a);
        R b = R(2) * dot(d, oc);
        R c = dot(oc, oc) - s.radius * s.radius;
        R sq = P::sqrt(P::madd(-(R(4) * c), b, b));   // b*b - 4*c
        R t = (-b - sq) / R(2);
        if (t < R(0.001)) t = (-b + sq) / R(2);
        if (t < R(0.001)) t = NaN();
        return along(o, t, d);
    }
    // Plane::intersect, src/geometry.cpp:99-105
    static V plane_intersect(const V& pt, const V& n, const V& o, const V& d) {
        V l2p = psub(pt, o);
        R t = dot(l2p, n) / dot(n, d);
        if (t < R(0.001)) t = NaN();
        if (t == INF() || t == -INF()) t = NaN();
        return along(o, t, d);
    }
    // Face::intersect, src/geometry.cpp:160-168
    static V face_intersect(const Prim& f, const V& o, const V& d) {
        V p = plane_intersect(f.a, f.n, o, d);
        int num = f.nv;
        for (int i = 0; i < num; i++) {
            V e = psub(f.verts[i], f.verts[(i + 1) % num]);
            R to_p = dot(e, psub(f.verts[i], p));
            R to_next = dot(e, psub(f.verts[i], f.verts[(i + 2) % num]));
            if (to_p * to_next < 0) return mk(NaN(), NaN(), NaN());
        }
        return p;
    }
    // Plane::get_material, src/geometry.cpp:106-116 (5-unit checker on world x, z)
    static int plane_material(const Prim& pl, const V& p) {
        V v = psub(p, pl.a);
        // (int) of an out-of-range double is undefined in C++; the reference's x86 build converts with cvttsd2si, which
        // returns INT_MIN for it (and for NaN): stated here so that the result does not depend on this file's compiler
        int x = cvttsd2si((double)(v.v[0] / R(5))) % 2;
        int z = cvttsd2si((double)(v.v[2] / R(5))) % 2;
        if (v.v[0] < 0) x = -x;
        if (v.v[2] < 0) z = -z;
        int m = (x == z) ? pl.mat0 : pl.mat1;
        if ((v.v[0] < 0) != (v.v[2] < 0)) m = (x == z) ? pl.mat1 : pl.mat0;
        return m;
    }

    // get_intersection, src/main.cpp:34-43.  Returns the index of the accepted
    // object or -1; p keeps (INF,INF,INF) when nothing was accepted.
    int closest(const V& o, const V& d, V* p, V* normal, int* mat) const {
        int hit = -1;
        for (int i = 0; i < (int)prims.size(); i++) {
            const Prim& pr = prims[i];
            V p0;
            if (pr.kind == KIND_SPHERE) p0 = sphere_intersect(pr, o, d);
            else if (pr.kind == KIND_PLANE) p0 = plane_intersect(pr.a, pr.n, o, d);
            else p0 = face_intersect(pr, o, d);
            if (valid(p0) && (length(psub(*p, o)) > length(psub(p0, o)))) {
                *p = p0;
                if (pr.kind == KIND_SPHERE) { *normal = normalize(psub(p0, pr.a)); *mat = pr.mat0; }
                else if (pr.kind == KIND_PLANE) { *normal = pr.n; *mat = plane_material(pr, p0); }
                else { *normal = pr.n; *mat = pr.mat0; }
                hit = i;
            }
        }
        return hit;
    }

    static R uniform(Rng& g) { return (R)g.next() / (R)RNG_MAX; }   // (double)g()/g.max()

    // Line::reflect, src/geometry.cpp:78-83: direction <- 2(n.d)n - d
    static V reflect_dir(const V& d, const V& n) {
        V w = vscale(n, dot(n, d));
        return vsub(vscale(w, R(2)), d);
    }

    // the rotation of a +z-hemisphere sample about the axis n, src/main.cpp:73,100
    static V rotate(const V& v, const V& n, R cost, R sint) {
        R nv = dot(n, v);
        V a = vscale(vsub(v, vscale(n, nv)), cost);
        V b = vscale(n, dot(n, v));
        V c = vscale(cross(n, v), sint);
        return vadd(vadd(a, b), c);
    }

    // get_reflection_ray, src/main.cpp:47-104.  Returns the new (un-normalised) direction.
    V reflection_dir(const V& d, const V& normal, const Mat& m, Rng& g) {
        const R PI = (R)M_PI;
        if (P::glossy(g.next(), m.gloss)) {
            V r = vscale(reflect_dir(d, normal), R(-1));
            R theta = P::acos(r.v[2]);
            V n = normalize(cross(mk(0, 0, 1), r));
            R sint = P::sin(theta), cost = P::cos(theta);
            for (long iter = 0;; iter++) {
                R u = P::pow(P::cos(uniform(g) * PI / R(2)), R(1.0) / m.sexp);
                R phi = uniform(g) * R(2) * PI;
                R sr = P::sqrt(R(1) - u * u);
                V v = mk(sr * P::cos(phi), sr * P::sin(phi), u);
                V nd;
                if (std::fabs(normal.v[2] - R(1)) < R(0.0001)) nd = v;
                else nd = rotate(v, n, cost, sint);
                if (dot(nd, normal) >= 0) return nd;
                if (iter > 100000000L) { guard_tripped = true; return nd; }  // the reference would spin forever
            }
        } else {
            R theta = P::acos(normal.v[2]);
            V n = normalize(cross(mk(0, 0, 1), normal));
            R sint = P::sin(theta), cost = P::cos(theta);
            R u = uniform(g);
            R phi = uniform(g) * R(2) * PI;
            R sr = P::sqrt(R(1) - u * u);
            V v = mk(sr * P::cos(phi), sr * P::sin(phi), u);
            if (std::fabs(normal.v[2] - R(1)) < R(0.0001)) return v;
            return rotate(v, n, cost, sint);
        }
    }

    // get_refraction_ray, src/main.cpp:107-122 with Line::reflect / Line::refract,
    // src/geometry.cpp:78-92.  Returns the new direction.
    static V refraction_dir(const V& d, const V& normal, const Mat& m, R n_cur) {
        R n2;
        V an = normal;
        if (std::fabs(m.rindex - n_cur) < R(0.0001)) { n2 = R(1); an = vscale(normal, R(-1)); }
        else n2 = m.rindex;
        R nv = dot(an, d);
        if ((n2 * n2) / (n_cur * n_cur) < R(1) - nv * nv) return reflect_dir(d, an);
        R ratio = n_cur / n2;
        R anv = std::fabs(dot(an, d));
        R k = ratio * anv - P::sqrt(R(1) - ratio * ratio * (R(1) - anv * anv));
        return normalize(vadd(vscale(d, ratio), vscale(an, k)));
    }

    // draws consumed by rand_point of the first light, src/geometry.cpp:120,145-150,175-179
    void light_draws(Rng& g) const {
        const Prim& l = prims[first_light];
        if (l.kind == KIND_SPHERE || l.kind == KIND_FACE) { g.next(); g.next(); }
    }

    // add_to_path + Path::trace for one path: src/main.cpp:126-183, src/geometry.cpp:186-217.
    // *first_hit receives the index of the object the camera ray hit, or -1.
    void trace(V o, V d, Rng& g, long* rays, int* first_hit, R out[3]) {
        std::vector<Vertex> st;
        int remaining = depth;
        R n_cur = R(1);
        V term = mk(INF(), INF(), INF());
        int term_mat = -1;
        bool lit = false;
        for (long seg = 0;; seg++) {
            (*rays)++;
            V p = mk(INF(), INF(), INF()), normal = mk(0, 0, 0);
            int mi = -1;
            int hit = closest(o, d, &p, &normal, &mi);
            if (seg == 0 && first_hit) *first_hit = hit;
            if (log) std::fprintf(log, "  seg %ld o (%.17g %.17g %.17g) d (%.17g %.17g %.17g) hit %d p (%.17g %.17g %.17g) n.d %.6g mat %d rindex %g n_cur %g remaining %d\n",
                                  seg, (double)o.v[0], (double)o.v[1], (double)o.v[2], (double)d.v[0], (double)d.v[1], (double)d.v[2], hit,
                                  (double)p.v[0], (double)p.v[1], (double)p.v[2], (double)dot(normal, d), mi, mi >= 0 ? (double)mats[mi].rindex : -1.0,
                                  (double)n_cur, remaining);
            if (mi >= 0 && mats[mi].is_light) { term = p; term_mat = mi; lit = true; break; }   // main.cpp:132-135
            if (!valid(p)) break;                                                                // main.cpp:179-182
            if (remaining == 0 && first_light >= 0) { light_draws(g); break; }                   // main.cpp:137-153 (always black)
            if (seg > 100000) { guard_tripped = true; break; }
            const Mat& m = mats[mi];
            V fwd; int change = 1; R n2 = R(1); bool reflect = false;
            if (std::fabs(m.rindex - n_cur) < R(0.0001)) {                  // leaving, main.cpp:158-161
                fwd = refraction_dir(d, normal, m, n_cur);
                change = 0;
            } else if (m.rindex == 0 || g.next() < RNG_MAX / 2) {           // main.cpp:164-167
                fwd = reflection_dir(d, normal, m, g);
                reflect = true;
            } else {                                                        // main.cpp:168-171
                fwd = refraction_dir(d, normal, m, n_cur);
                n2 = m.rindex;
            }
            Vertex vx;
            vx.p = p; vx.n = normal; vx.r = vscale(reflect_dir(d, normal), R(-1)); vx.mat = mi; vx.reflect = reflect;   // main.cpp:173-176
            st.push_back(vx);
            o = p; d = fwd; remaining -= change; n_cur = n2;
        }
        out[0] = out[1] = out[2] = R(0);
        if (!lit || !valid(term)) return;                                   // geometry.cpp:189
        // Path::trace, geometry.cpp:186-217: from the light back to the camera
        R c[3] = {mats[term_mat].ref[0], mats[term_mat].ref[1], mats[term_mat].ref[2]};
        V next = term;
        for (int k = (int)st.size() - 1; k >= 0; k--) {
            const Vertex& vx = st[k];
            const Mat& m = mats[vx.mat];
            V w = normalize(psub(next, vx.p));
            const R* dc; const R* sc; R s;
            if (vx.reflect) { dc = m.diff; sc = m.spec; s = P::pow(dot(vx.r, w), m.sexp); }
            else { dc = m.rdiff; sc = m.rspec; s = R(1); }
            R dd = dot(vx.n, w);
            if (s < 0) s = 0;
            if (dd < 0) dd = 0;
            for (int i = 0; i < 3; i++) c[i] = c[i] * (dc[i] * dd + sc[i] * s);
            next = vx.p;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
    }

    // one pixel: the body of do_rays_i, src/main.cpp:252-275
    void pixel(int x, int y, Rng& g, long* rays, R out[3], int32_t* hit_ids) {
        const V camera = mk(cam[0], cam[1], cam[2]);                        // main.cpp:16: (0,0,40) unless the header moves it
        const R PI = (R)M_PI;
        R acc[3] = {0, 0, 0};
        for (int i = 0; i < spp; i++) {
            R xs = uniform(g) - R(0.5);
            R ys = uniform(g) - R(0.5);
            V px = mk((R)(x - width / 2 + xs) / scale, (R)(height / 2 - y + ys) / scale, 0);
            R theta = uniform(g) * R(2) * PI;
            R rad = uniform(g) * lens;
            V ls = mk(rad * P::cos(theta), rad * P::sin(theta), 0);
            V D = normalize(psub(px, camera));
            V C = along(camera, focal, D);
            V org = mk(camera.v[0] + ls.v[0], camera.v[1] + ls.v[1], camera.v[2] + ls.v[2]);
            V dir = normalize(psub(C, org));
            R c[3]; int fh = -1;
            trace(org, dir, g, rays, &fh, c);
            if (hit_ids) hit_ids[i] = fh;
            for (int k = 0; k < 3; k++) acc[k] = acc[k] + c[k];
        }
        for (int k = 0; k < 3; k++) out[k] = acc[k] / (R)spp;
    }
};

// Renders rows [row_begin,row_end) of the frame (all rows for STREAM mode, which
// ignores the range) with n_workers CPU threads.
//   rgb      : width*height*3 doubles, un-exposed radiance (float results widened)
//   rays     : STREAM: num_threads counters (one per reference thread); PIXEL: 1 counter
//   hit_ids  : optional, width*height*spp int32 (object index hit by each camera ray, -1 = none)
//   pix_rays : optional, width*height int32 (rays spent on each pixel)
template <class P>
int render(const SceneD& sc, const double* params, int rng_mode, int n_workers, int row_begin, int row_end,
           double* rgb, int64_t* rays, int32_t* hit_ids, int32_t* pix_rays) {
    Tracer<P> base;
    base.load(sc, params);
    const int W = base.width, H = base.height, spp = base.spp;
    int T = (rng_mode == RNG_STREAM) ? (int)params[PA_THREADS] : n_workers;
    if (T < 1) T = 1;
    std::vector<int> tripped(T, 0);
    std::vector<int64_t> cnt(T, 0);
    auto work = [&](int t) {
        Tracer<P> tr = base;
        typename P::Real out[3];
        long r = 0;
        if (rng_mode == RNG_STREAM) {
            Rng g;                                                         // state 1, main.cpp:301,307
            for (int y = t; y < H; y += T)                                 // main.cpp:250
                for (int x = 0; x < W; x++) {
                    long r0 = r;
                    tr.pixel(x, y, g, &r, out, hit_ids ? hit_ids + ((size_t)y * W + x) * spp : nullptr);
                    for (int k = 0; k < 3; k++) rgb[((size_t)y * W + x) * 3 + k] = (double)out[k];
                    if (pix_rays) pix_rays[(size_t)y * W + x] = (int32_t)(r - r0);
                }
        } else {
            for (int y = row_begin + t; y < row_end; y += T)
                for (int x = 0; x < W; x++) {
                    Rng g; g.s = pixel_seed((uint32_t)(y * W + x), base.frame_seed);
                    long r0 = r;
                    tr.pixel(x, y, g, &r, out, hit_ids ? hit_ids + ((size_t)y * W + x) * spp : nullptr);
                    for (int k = 0; k < 3; k++) rgb[((size_t)y * W + x) * 3 + k] = (double)out[k];
                    if (pix_rays) pix_rays[(size_t)y * W + x] = (int32_t)(r - r0);
                }
        }
        cnt[t] = r;
        tripped[t] = tr.guard_tripped;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    int bad = 0;
    if (rng_mode == RNG_STREAM) for (int t = 0; t < T; t++) rays[t] = cnt[t];
    else { int64_t s = 0; for (int t = 0; t < T; t++) s += cnt[t]; rays[0] = s; }
    for (int t = 0; t < T; t++) bad |= tripped[t];
    return bad ? -1 : 0;
}

// expose + blur + Color::to_int: src/main.cpp:186-242, src/geometry.cpp:15-23.
// Always double, as the reference; rgb is modified in place (bloom added).
static inline void expose(int W, int H, double exposure, int num_blurs, int n_workers, double* rgb, int32_t* img) {
    const int radius = 20; const double stddev = 3;
    const int K = 2 * radius + 1;
    std::vector<double> kernel((size_t)K * K);
    double denom = 2 * stddev * stddev;
    for (int y = -radius; y <= radius; y++)
        for (int x = -radius; x <= radius; x++)
            kernel[(size_t)(y + radius) * K + x + radius] = std::exp(-(x * x + y * y) / denom) / denom / M_PI;
    std::vector<double> cur((size_t)W * H * 3), nxt;
    for (size_t i = 0; i < (size_t)W * H * 3; i++) cur[i] = (rgb[i] * exposure > 1) ? exposure : 0.0;
    for (int b = 1; b < 1 + num_blurs; b++) {
        nxt.assign((size_t)W * H * 3, 0.0);
        auto rows = [&](int t, int T) {
            for (int y = t; y < H; y += T)
                for (int x = 0; x < W; x++) {
                    double c[3] = {0, 0, 0};
                    for (int ky = -radius; ky <= radius; ky++)
                        for (int kx = -radius; kx <= radius; kx++)
                            if (y + ky >= 0 && y + ky < H && x + kx >= 0 && x + kx < W) {
                                const double* s = &cur[((size_t)(y + ky) * W + (x + kx)) * 3];
                                double k = kernel[(size_t)(ky + radius) * K + kx + radius];
                                for (int i = 0; i < 3; i++) c[i] = c[i] + s[i] * k;
                            }
                    for (int i = 0; i < 3; i++) nxt[((size_t)y * W + x) * 3 + i] = c[i];
                }
        };
        int T = n_workers < 1 ? 1 : n_workers;
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(rows, t, T);
        rows(0, T);
        for (auto& x : th) x.join();
        cur.swap(nxt);
    }
    for (size_t i = 0; i < (size_t)W * H * 3; i++) rgb[i] = rgb[i] + cur[i];
    double mx = 1.0 / exposure;
    for (size_t i = 0; i < (size_t)W * H; i++) {
        int r = cvttsd2si(255 * rgb[3 * i] / mx), g = cvttsd2si(255 * rgb[3 * i + 1] / mx), b = cvttsd2si(255 * rgb[3 * i + 2] / mx);
        if (r > 255) r = 255;
        if (g > 255) g = 255;
        if (b > 255) b = 255;
        img[i] = (r << 16) + (g << 8) + b;
    }
}

}  // namespace rto
