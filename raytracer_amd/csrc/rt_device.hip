// librt_hip.so -- render kernels for gfx950 (MI355X) and their C ABI (include/rt_hip.h).
//
// What is computed: the per-pixel render loop of samkottler/rayTracer,
// /root/reference/src/main.cpp:247-289 (do_rays_i) with everything it calls:
// add_to_path :126-183, get_intersection :34-43, get_reflection_ray :47-104,
// get_refraction_ray :107-122, the primitives of /root/reference/src/geometry.cpp
// :94-179 and the shading product Path::trace :186-217.
//
// How it is organised on the GPU:
//   * one lane owns one pixel at a time and runs its paths_per_pixel paths in
//     order from a per-pixel minstd_rand0 stream (the reference's generator,
//     seeded per pixel instead of per CPU thread);
//   * lanes are persistent: a lane whose pixel is finished takes the next pixel
//     from a global queue (one ballot + one atomic per wave, the new work packed
//     into the idle lanes by their rank in the ballot), and a lane whose path ended
//     starts its next camera ray at once -- so every lane that enters the
//     closest-hit scan carries a live ray, whatever its neighbours' path lengths;
//   * closest hit (get_intersection) is split in two:
//       broad phase, FP32, the hot loop: every primitive has a bounding sphere
//         (a sphere is its own; a face gets the sphere around its vertices; a plane
//         an infinite one) stored as one float4 in LDS, in file order.  All lanes
//         of a wave read the same record (LDS broadcast) and evaluate a
//         conservative ray/sphere discriminant in 10 VALU instructions; the error
//         bound of the float evaluation is folded into per-ray and per-record
//         constants, so a primitive the exact test could accept is never dropped.
//         Survivors (a handful per ray) are appended to a per-lane candidate list;
//       narrow phase, FP64: the candidates are evaluated in file order with the
//         reference's own double-precision operation sequence (no fused
//         multiply-add, IEEE divide and sqrt), including its acceptance rule on
//         Euclidean distance and its tie order;
//   * lists larger than the LDS window are streamed through it in chunks with the
//     workgroup in lock-step; smaller lists are staged once and the waves run free;
//   * the reference's recursion and linked vertex list become a loop and a
//     per-lane stack of per-vertex throughput factors, multiplied light-to-camera
//     at the end exactly as Path::trace does.
//
// Arithmetic contract: outside the broad phase every value is a double and every
// operation is the reference's, rounded separately (-ffp-contract=off), so results
// are reproducible bit for bit on a CPU; only sin/cos/acos/pow differ from the
// reference binary (rt_math.h instead of glibc).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_math.h"

namespace {

// ------------------------------------------------------------------ device data
constexpr double RT_EPS = 0.001;               // Sphere/Plane::intersect, geometry.cpp:102,135-136
constexpr int RT_MAXV = 2 * RT_MAX_DEPTH + 2;  // bounce-stack entries per lane
constexpr int BROAD_UNROLL = 4;
constexpr int CAND_CAP = 48;                   // candidate list entries per lane before an early narrow phase
constexpr unsigned ERR_REJECT_GUARD = 1u, ERR_STACK = 2u;
constexpr int REJECT_GUARD_ITERS = 1 << 20;
constexpr float F32_U = 5.9604645e-8f;         // 2^-24, unit roundoff of float

struct ChunkDesc { int gofs, count, first_index, pad; };   // count: records incl. NaN padding to a multiple of 32

// The exact phase reads these tables lane by lane at unrelated addresses (each lane its own solid or material), and
// the vector L1 then handles one request per lane and instruction whatever its width.  The records are therefore
// aligned and ordered so that what a test needs comes in few 16-byte loads (ld2 / ldi4 below).
struct alignas(32) PrimD {   // 96 bytes, indexed by file order: the narrow phase's exact data
    double a[3];             // sphere centre | plane point | face verts[0]
    double radius;
    int kind, mat0, mat1, voff;
    int vcnt, pad0, pad1, pad2;
    double n[3];             // plane / face unit normal
    double pad3;
};

// One face vertex with the two quantities Face::intersect (geometry.cpp:162-166) derives from the vertex
// list for every ray: e = v_i - v_i+1 and to_next = e.(v_i - v_i+2), computed once on the host with the
// same double operations.
struct alignas(32) FaceV { double v[3]; double e[3]; double to_next; double pad; };

struct alignas(16) MatD {    // struct Material (geometry.hpp:18-29) in double, 144 bytes
    unsigned is_light;
    unsigned gloss_thr;      // glossy iff rng < gloss_thr  (= ceil(gloss * 2147483646), main.cpp:48)
    double rindex;           // the three above: what a hit decides on, one 16-byte load
    double sexp;
    double ref[3];
    double diff[3], spec[3];     // 48 bytes from offset 48
    double rdiff[3], rspec[3];   // 48 bytes from offset 96
};

struct KParams {
    int width, height, spp, depth;
    double scale, lens, focal;
    double cam[3];           // camera, main.cpp:16
    unsigned frame_seed;
    int has_light, light_draws;
    int tile_rows, rank, world;
    int stream_mode, n_streams;
    unsigned total_items;
    int nchunks;
    int n_prims;
    int max_verts;           // bounce-stack entries available per path (<= RT_MAXV)
    float cmax;              // max |C - T| + R over finite bounding spheres
    double T[3];             // translation applied before rounding coordinates to float
    const float4* records;
    const ChunkDesc* chunks;
    const PrimD* prims;
    const FaceV* verts;
    const MatD* mats;
    const double* always;    // solids every ray must test exactly (unbounded or very large bounding sphere): per solid
                             // ALW_HEAD doubles (kind, file index, vertex count, -, point[3], radius, normal[3], -) and
                             // 8 doubles per face vertex (FaceV); their broad-phase records are padding
    int n_always, always_doubles;
    double* rgb;
    int* hit_ids;
    unsigned long long* stream_rays;
    unsigned* ctrl;          // [0] queue head, [1] error bits, [2..3] rays (u64), [4..5] candidates (u64), [6] rays whose candidate list overflowed
};

struct d3 { double x, y, z; };

#define DEV __device__ __forceinline__

DEV d3 dmk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV d3 dsub(d3 a, d3 b) { return dmk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV d3 dadd(d3 a, d3 b) { return dmk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV d3 dscale(d3 a, double s) { return dmk(s * a.x, s * a.y, s * a.z); }
DEV d3 dneg(d3 a) { return dmk(-a.x, -a.y, -a.z); }
// Vector::dot, Vector.cpp:33-39: total = 0; total += a[i]*b[i], every step rounded
DEV double ddot(d3 a, d3 b) {
    double t = 0.0;
    t = t + a.x * b.x;
    t = t + a.y * b.y;
    t = t + a.z * b.z;
    return t;
}
DEV double dlen(d3 a) { return __builtin_sqrt(ddot(a, a)); }
DEV d3 dnorm(d3 a) { double l = dlen(a); return dmk(a.x / l, a.y / l, a.z / l); }
DEV d3 dcross(d3 a, d3 b) {
    return dmk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Point + t*direction, geometry.cpp:55-57 after Vector.cpp:67-69: product, then sum
DEV d3 dalong(d3 o, double t, d3 d) { return dmk(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); }
DEV bool dnan(double x) { return x != x; }
// Point::is_valid, geometry.cpp:51-54
DEV bool dvalid(d3 p) {
    const double inf = __builtin_inf();
    return !(dnan(p.x) || dnan(p.y) || dnan(p.z) || p.x == inf || p.y == inf || p.z == inf);
}

// 16-byte loads and stores of table records (addresses 16-byte aligned by the records' layout)
DEV double2 ld2(const double* p) { return *(const double2*)p; }
DEV void st2(double* p, double a, double b) { *(double2*)p = make_double2(a, b); }
DEV int4 ldi4(const int* p) { return *(const int4*)p; }

// minstd_rand0: x <- 16807 x mod (2^31 - 1), by folding the 46-bit product (2^31 == 1 mod m)
DEV unsigned rng_next(unsigned& s) {
    unsigned long long p = (unsigned long long)s * 16807ull;
    unsigned x = (unsigned)(p & 0x7fffffffull) + (unsigned)(p >> 31);
    if (x >= 0x7fffffffu) x -= 0x7fffffffu;
    s = x;
    return x;
}
// (double)g() / g.max(), main.cpp:255
DEV double rng_uniform(unsigned& s) { return (double)rng_next(s) / 2147483646.0; }

__host__ __device__ inline unsigned pixel_seed(unsigned pixel_index, unsigned frame_seed) {
    unsigned h = pixel_index * 0x9E3779B1u ^ frame_seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return 1u + (h % 2147483646u);
}

struct Hit {
    double q;       // |p - o|^2 (as the reference rounds it, before its square root) of the closest accepted hit so far
    int idx;        // file index of its solid, -1 = none
    d3 p;
};

// get_intersection's acceptance, main.cpp:37: scanning in file order and taking a candidate iff its point is
// valid and strictly nearer selects the smallest (distance, file index) pair.  Written as that pair
// comparison, the selection no longer depends on the order in which candidates are examined.
// The reference compares len = sqrt(q).  The correctly rounded square root is monotonic, and two arguments
// that differ by more than 2^-50 relative have roots that differ by more than 2^-51 relative -- more than
// the two roundings (2^-53 each) can close -- so outside that band the squared lengths decide and the square
// roots (the most expensive instructions of the test) are only taken inside it.
DEV void consider_q(double q, d3 p, int idx, Hit& h) {
    const double band = 8.8817841970012523e-16;      // 2^-50
    bool take;
    if (q < h.q * (1.0 - band)) take = true;
    else if (q > h.q * (1.0 + band)) take = false;
    else {
        const double len = __builtin_sqrt(q), hl = __builtin_sqrt(h.q);
        take = hl > len || (hl == len && idx < h.idx);
    }
    if (take) { h.q = q; h.idx = idx; h.p = p; }
}
// |p - o|^2 of a valid point; NaN marks an invalid one (a valid point is finite or -inf per component and the
// origin of a ray that produced it is finite, so the sum of squares is never NaN)
DEV double hit_q(d3 p, d3 o) {
    if (!dvalid(p)) return __builtin_nan("");
    const d3 v = dsub(p, o);
    return ddot(v, v);
}
DEV void consider(d3 p, int idx, d3 o, Hit& h) {
    const double q = hit_q(p, o);
    if (q == q) consider_q(q, p, idx, h);
}

// Sphere::intersect, geometry.cpp:129-138
DEV d3 sphere_point(d3 centre, double radius, d3 o, d3 d) {
    d3 oc = dsub(o, centre);
    double b = 2.0 * ddot(d, oc);
    double c = ddot(oc, oc) - radius * radius;
    // Origin outside and centre behind: with b > 0 and c > 0 the rounded sqrt(b*b - 4c) cannot exceed b
    // (sqrt(fl(b*b)) rounds to b), so both roots fail t >= 0.001 and the reference returns a NaN point too.
    if (b > 0.0 && c > 0.0) return dmk(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
    double sq = __builtin_sqrt(b * b - 4.0 * c);
    double t = (-b - sq) / 2.0;
    if (t < RT_EPS) t = (-b + sq) / 2.0;
    if (t < RT_EPS) t = __builtin_nan("");
    return dalong(o, t, d);
}
// Plane::intersect, geometry.cpp:99-105
DEV d3 plane_point(d3 P, d3 n, d3 o, d3 d) {
    const double num = ddot(dsub(P, o), n), den = ddot(n, d);
    // a correctly rounded quotient keeps the sign of the exact one: opposite signs give t < 0 < 0.001
    if ((num < 0.0 && den > 0.0) || (num > 0.0 && den < 0.0)) return dmk(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
    double t = num / den;
    const double inf = __builtin_inf();
    if (t < RT_EPS) t = __builtin_nan("");
    if (t == inf || t == -inf) t = __builtin_nan("");
    return dalong(o, t, d);
}
// Face::intersect, geometry.cpp:160-168
DEV d3 face_point(d3 a, d3 n, int num, const FaceV* v, d3 o, d3 d) {
    d3 p = plane_point(a, n, o, d);
    if (dnan(p.x)) return p;                 // plane missed: every comparison below would be false anyway
    bool miss = false;
    for (int i = 0; i < num; i++) {
        const double* f = v[i].v;
        const double2 q0 = ld2(f), q1 = ld2(f + 2), q2 = ld2(f + 4), q3 = ld2(f + 6);   // v0 v1 | v2 e0 | e1 e2 | to_next -
        double to_p = ddot(dmk(q1.y, q2.x, q2.y), dsub(dmk(q0.x, q0.y, q1.x), p));
        if (to_p * q3.x < 0.0) miss = true;
    }
    if (miss) return dmk(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
    return p;
}

// Line::reflect, geometry.cpp:78-83: 2 (n.d) n - d
DEV d3 reflect_dir(d3 d, d3 n) {
    d3 w = dscale(n, ddot(n, d));
    return dsub(dscale(w, 2.0), d);
}

// rotation of a +z hemisphere sample v about the unit axis n, main.cpp:73,100
DEV d3 rotate3(d3 v, d3 n, double cost, double sint) {
    double nv = ddot(n, v);
    d3 a = dscale(dsub(v, dscale(n, nv)), cost);
    d3 b = dscale(n, nv);
    d3 c = dscale(dcross(n, v), sint);
    return dadd(dadd(a, b), c);
}

// get_reflection_ray, main.cpp:47-104.  Returns the new direction (not re-normalised).  The glossy
// branch (:48-77, power-cosine lobe about the mirror direction, rejection loop) and the diffuse branch
// (:78-103, uniform hemisphere about the normal) build their sample the same way about different axes;
// they share one instruction stream here, each lane keeping its own branch's operation sequence.
DEV d3 reflection_dir(d3 d, d3 normal, unsigned gloss_thr, double sexp, unsigned& rng, unsigned* err, unsigned item) {
    const double PI = 3.14159265358979323846;
    const bool flat = __builtin_fabs(normal.z - 1.0) < 0.0001;
    const bool glossy = rng_next(rng) < gloss_thr;
    const d3 axis = glossy ? dneg(reflect_dir(d, normal)) : normal;
    const double theta = rtm::acosd(axis.z);
    const d3 n = dnorm(dcross(dmk(0.0, 0.0, 1.0), axis));
    double sint, cost;
    rtm::sincosd(theta, &sint, &cost);
    const double inv_exp = 1.0 / sexp;
    d3 nd = dmk(0.0, 0.0, 0.0);
    for (int iter = 0;; iter++) {
        const double u1 = rng_uniform(rng);
        const double u = glossy ? rtm::powd(rtm::cosd(u1 * PI / 2.0), inv_exp) : u1;
        const double phi = rng_uniform(rng) * 2.0 * PI;
        const double sr = __builtin_sqrt(1.0 - u * u);
        double sp, cp;
        rtm::sincosd(phi, &sp, &cp);
        const d3 v = dmk(sr * cp, sr * sp, u);
        nd = flat ? v : rotate3(v, n, cost, sint);
        if (!glossy || ddot(nd, normal) >= 0.0) break;
        if (iter > REJECT_GUARD_ITERS) { atomicOr(err, ERR_REJECT_GUARD); atomicMax(err + 6, item + 1u); break; }
    }
    return nd;
}

// get_refraction_ray, main.cpp:107-122, with Line::reflect / Line::refract, geometry.cpp:78-92
DEV d3 refraction_dir(d3 d, d3 normal, double rindex, double n_cur) {
    double n2;
    d3 an = normal;
    if (__builtin_fabs(rindex - n_cur) < 0.0001) { n2 = 1.0; an = dneg(normal); }
    else n2 = rindex;
    double nv = ddot(an, d);
    if ((n2 * n2) / (n_cur * n_cur) < 1.0 - nv * nv) return reflect_dir(d, an);
    double ratio = n_cur / n2;
    double anv = __builtin_fabs(nv);
    double k = ratio * anv - __builtin_sqrt(1.0 - ratio * ratio * (1.0 - anv * anv));
    return dnorm(dadd(dscale(d, ratio), dscale(an, k)));
}

// Plane::get_material, geometry.cpp:106-116
DEV int plane_material(double ax, double az, int mat0, int mat1, d3 p) {
    double vx = p.x - ax, vz = p.z - az;
    int x = (int)(vx / 5.0) % 2;
    int z = (int)(vz / 5.0) % 2;
    if (vx < 0.0) x = -x;
    if (vz < 0.0) z = -z;
    int m = (x == z) ? mat0 : mat1;
    if ((vx < 0.0) != (vz < 0.0)) m = (x == z) ? mat1 : mat0;
    return m;
}

// ---- broad phase ---------------------------------------------------------------------------
// Conservative float test of a ray against a bounding sphere (centre c, radius R), all
// coordinates relative to the translation T and rounded to float.  With
//     hb = d.(o - c) = d.o - d.c            oo = |o - c|^2 - R^2 = |o|^2 + (|c|^2 - R^2) - 2 o.c
// the exact discriminant is hb^2 - oo.  The kernel evaluates, in 8 VALU instructions per record,
//     hb' = od' - d'.c        (d' = d sqrt(1+delta), od' = d'.o: three fma)
//     nq  = (R^2 - |c|^2) + 2 o.c            (three fma, the record carries R^2 - |c|^2)
//     s   = hb'^2 + nq                        (one fma)          keep iff  s >= thr  (one compare)
// with thr = |o|^2 - E0 - Eh^2 (1 + 1/delta).  If the float evaluation of hb is off by at most Eh
// and that of oo by at most E0, then  hb^2 <= (|hb~| + Eh)^2 <= hb~^2 (1+delta) + Eh^2 (1+1/delta)
// and  oo >= oo~ - E0, so a non-negative exact discriminant implies s >= thr: no primitive the
// exact test could accept is ever dropped.  Bounds (S = |o| + cmax, u = 2^-24), with a factor 2 of
// slack over the term-by-term count:  Eh = 14 u S,  E0 = 24 u S^2.
// A kept sphere is still dropped if it lies wholly behind the origin: hb~ > Eh and oo~ > E0.
constexpr float BROAD_DELTA = 1.9e-6f;         // used in thr; the applied scale 1.000001^2 - 1 is larger
constexpr float BROAD_SCALE = 1.000001f;

struct RayRec {
    float4 a;   // -d' (xyz), od'
    float4 b;   // 2 o (xyz), thr
    float4 c;   // behind thresholds: hb' > c.x and nq < c.y ; zw unused
};

DEV RayRec broad_setup(d3 o, d3 d, const double* T, float cmax) {
    RayRec r;
    const float ox = (float)(o.x - T[0]), oy = (float)(o.y - T[1]), oz = (float)(o.z - T[2]);
    const float fx = (float)d.x * BROAD_SCALE, fy = (float)d.y * BROAD_SCALE, fz = (float)d.z * BROAD_SCALE;
    const double od = (double)fx * ox + (double)fy * oy + (double)fz * oz;
    const double o2 = (double)ox * ox + (double)oy * oy + (double)oz * oz;
    const float S = __builtin_sqrtf((float)o2) * 1.0001f + cmax;
    const float Eh = 14.0f * F32_U * S;
    const float E0 = 24.0f * F32_U * S * S;
    r.a = make_float4(-fx, -fy, -fz, (float)od);
    const double thr = o2 - (double)E0 - (double)Eh * Eh * (1.0 + 1.0 / (double)BROAD_DELTA);
    // rounded down, and strictly below: the pipeline keeps a record iff thr - s is negative
    const float thr_f = (float)thr - __builtin_fabsf((float)thr) * (4.0f * F32_U) - 1e-30f;
    r.b = make_float4(2.0f * ox, 2.0f * oy, 2.0f * oz, thr_f);
    const float bc = (float)(o2 - (double)E0) - __builtin_fabsf((float)o2) * (4.0f * F32_U);
    r.c = make_float4(Eh * 1.00001f, bc, 0.0f, 0.0f);
    return r;
}

// s and hb' of one record (cx, cy, cz, R^2 - |c|^2) for one ray; nq is returned for the behind test
DEV float broad_s(const float4 rec, const float4 ra, const float4 rb, float& hb, float& nq) {
    hb = __builtin_fmaf(ra.z, rec.z, __builtin_fmaf(ra.y, rec.y, __builtin_fmaf(ra.x, rec.x, ra.w)));
    nq = __builtin_fmaf(rb.z, rec.z, __builtin_fmaf(rb.y, rec.y, __builtin_fmaf(rb.x, rec.x, rec.w)));
    return __builtin_fmaf(hb, hb, nq);
}
DEV bool broad_behind(float hb, float nq, const float4 rc) { return hb > rc.x && nq < rc.y; }

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
        narrow_always(kp.always, kp.n_always, o, d, h);
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
            if (!segment(s, h, fst, item, item_rays, kp)) state = ST_NEED;
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

// ---- three-kernel pipeline (per-pixel mode) ----------------------------------------------------
// Path state lives in global memory, one slot per pixel of the batch, structure-of-arrays.  A frame is
//   init_kernel                     camera ray of every pixel, queue 0 = all slots
//   repeat: broad_kernel            FP32 scan of the LDS-resident records for every queued ray
//           shade_kernel            FP64 narrow phase + segment(); surviving rays are packed into the
//                                   next queue by ballot rank (one atomic per wave)
// until a queue is empty.  Each kernel gets its own register allocation: the hot broad_kernel stays
// small and runs at high occupancy, the branchy double-precision shade_kernel may use what it needs.
#ifndef SHADE_PROFILE
#define SHADE_PROFILE 0
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
    uint2* cand;               // [WF_CAND_CAP][slots]: entry e of the ray at queue position pos is cand[e * slots + pos]
                               // (neighbouring lanes read neighbouring words): (word index j, keep mask of records 32j..32j+31,
                               // bit 31-k = record 32j+k), only the non-zero words, ascending j
    unsigned* cand_cnt;        // [2][slots] (one per queue: a round's shade_kernel reads one and clears the other) entries used; above WF_CAND_CAP (WF_OVERFLOW as a rule): more non-zero words than fit
};

// the ray itself (o, d) was read before the closest-hit search and is passed in
DEV void load_state(const WfParams& w, int q, unsigned pos, unsigned slot, d3 ro, d3 rd, PathState& s) {
    const size_t n = w.slots;
    const int* si = w.si[q] + pos;
    s.rng = (unsigned)si[SI_RNG * n];
    s.sample = si[SI_SAMPLE * n];
    const int pk = si[SI_PACK * n];
    s.pv_mat = si[SI_PVMAT * n];
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
    const RayRec r = broad_setup(s.o, s.d, w.kp.T, w.kp.cmax);
    w.rec[0][pos] = r.a; w.rec[1][pos] = r.b;
    w.cand_cnt[(size_t)q * w.slots + pos] = 0u;        // broad_kernel's slices add to it
}

__global__ __launch_bounds__(256) void init_kernel(const WfParams w) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned slot = blockIdx.x * blockDim.x + threadIdx.x; slot < w.slots; slot += stride) {
        PathState s;
        const unsigned item = w.item_base + slot;
        map_item_to_pixel(item, w.kp, s.px, s.py);
        s.rng = pixel_seed((unsigned)(s.py * w.kp.width + s.px), w.kp.frame_seed);
        s.pv_mat = 0; s.pv_n = dmk(0, 0, 0); s.pv_r = dmk(0, 0, 0);
        s.acc_mem = w.acc + slot; s.acc_stride = w.slots;
        begin_pixel(s, item, w.kp);
        emit_ray(w, slot, slot, s, 0);
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
        const float S = __builtin_sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]) * 1.0001f + w.kp.cmax;
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
                const float M = (48.0f * F32_U) * __builtin_fmaf(S, __builtin_sqrtf(oc2), oc2) + (16.0f * F32_U) * __builtin_fmaf(hb, hb, oc2) + 1e-12f;
                if (disc > M) {
                    const float sq = __builtin_sqrtf(disc);
                    const float Mt = 2.0f * __builtin_sqrtf(M) + (8.0f * F32_U) * (__builtin_fabsf(hb) + sq + S) + 1e-6f;
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
                test_item(make_float4(__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w)), c1.x, __uint_as_float(c1.y));
                for (unsigned i = 0; i + 1u < cnt; i++) {
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

// Closest hit of the ray queued at `pos` (get_intersection, main.cpp:34-43): the always-tested solids (table `alw`, LDS)
// and the ray's kept records through the reference's exact test.  Called by whole waves.
//
// A ray that keeps very many records (a NaN ray keeps every record, a ray with a distant origin has wide error bounds;
// list overflow is the extreme case) would hold one lane busy with hundreds of exact tests while its wave, its workgroup
// (barrier) and in the end the kernel wait: 1000 tests are 0.25 ms, more than a whole round of a small frame.  Such a
// ray is tested against EVERY solid by its whole wave instead -- lane l takes solids l, l+64, ... -- and the 64 partial
// results are merged with get_intersection's order-independent acceptance (consider_q).
constexpr unsigned HEAVY_WORDS = 6, HEAVY_BITS = 48;   // shared out: lists of more than 6 mask words with more than 48 records

DEV Hit find_hit(const WfParams& w, unsigned pos, bool live, d3 o, d3 d, const double* alw, unsigned long long& ncand_total) {
    const int lane = threadIdx.x & 63;
    Hit h;
    h.q = __builtin_inf(); h.idx = -1;
    h.p = dmk(__builtin_inf(), __builtin_inf(), __builtin_inf());
    const unsigned n = live ? w.cand_cnt[(size_t)(w.iter & 1) * w.slots + pos] : 0u;
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
        narrow_always(alw, w.kp.n_always, o, d, h);
        ncand_total += (unsigned)w.kp.n_always;
        for (unsigned e = 0; e < n; e++) {
            const uint2 ent = cur;
            cur = nxt;
            if (e + 2u < n) nxt = cl[(e + 2u) * cstride];
            unsigned m = ent.y;
            ncand_total += (unsigned)__popc(m);
            while (m) {
                const int k = __clz((int)m);
                m &= ~(0x80000000u >> k);
                const int idx = (int)ent.x * BROAD_GROUP + k;
                // a ray with a NaN direction makes every s NaN and thr - NaN negative, padding records
                // included: indices past the list are never dereferenced
                if (idx < w.kp.n_prims) narrow_one(idx, o, d, w.kp, h);
            }
        }
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

// exact narrow phase + one segment of add_to_path per queued ray; survivors are packed into the next queue
__global__ __launch_bounds__(SHADE_BLOCK, SHADE_WAVES * 256 / SHADE_BLOCK) void shade_kernel(const WfParams w) {
    extern __shared__ double alw[];
    const unsigned count = w.counts[w.iter];
    if (count == 0) return;
    for (int i = threadIdx.x; i < w.kp.always_doubles; i += SHADE_BLOCK) alw[i] = w.kp.always[i];
    __syncthreads();
#if SHADE_PROFILE
    // wave-cycles per section (find_hit | segment | barriers | emit), summed over waves into ctrl[16..]
    unsigned long long prof[5] = {0, 0, 0, 0, 0};
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
        bool alive = false;
        PathState s;
        unsigned slot = 0;
        SHADE_TICK(0);
        // the closest hit needs the ray only; the rest of the path state is loaded after it (fewer live registers)
        d3 ro = dmk(0, 0, 0), rd = dmk(0, 0, 1);
        if (live) {
            const double2* sd = w.sd[qi] + pos;
            const size_t n = w.slots;
            const double2 p0 = sd[0], p1 = sd[n], p2 = sd[2 * n];
            ro = dmk(p0.x, p0.y, p1.x);
            rd = dmk(p1.y, p2.x, p2.y);
        }
        const Hit h = find_hit(w, pos, live, ro, rd, alw, ncand_total);   // whole waves: heavy rays are shared out
        SHADE_TICK(1);
        if (live) {
            slot = w.queue[qi][pos];
            load_state(w, qi, pos, slot, ro, rd, s);
            nrays++;
            FactorStackV fst; fst.base = w.stack + (size_t)slot * 4u * (unsigned)w.kp.max_verts;
            alive = segment(s, h, fst, w.item_base + slot, 0ull, w.kp);
        }
        SHADE_TICK(2);
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
        if (alive) emit_ray(w, base + rank, slot, s, qo);
        SHADE_TICK(4);
    }
#if SHADE_PROFILE
    if (lane == 0) for (int k = 0; k < 5; k++) atomicAdd((unsigned long long*)(w.kp.ctrl + 16 + 2 * k), prof[k]);
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

// ---- expose(): bloom + 8-bit pack (main.cpp:186-242, geometry.cpp:15-23) -------------------------
// Everything in double, every product and sum rounded on its own, taps in the reference's order (ky outer, kx
// inner), so the packed frame is the reference's bit for bit.  A tap outside the frame is skipped by the
// reference; here it reads +0.0 from the zero-filled border of the tile, and acc + (+0.0 * k) == acc because
// acc never is -0.0 (it starts at +0.0 and a sum of doubles is -0.0 only when both terms are).
constexpr int BLOOM_R = 20, BLOOM_K = 2 * BLOOM_R + 1, BLOOM_T = 16, BLOOM_TILE = BLOOM_T + 2 * BLOOM_R;

// B0 of main.cpp:198-205: exposure where the channel is over-exposed, else 0
__global__ void expose_mask_kernel(const double* rgb, double* cur, size_t n3, double exposure) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x)
        cur[i] = (rgb[i] * exposure > 1.0) ? exposure : 0.0;
}

// one pass of main.cpp:207-223 for one channel (blockIdx.z) of a 16x16 tile of pixels
// block_rows: null = every 16-row block of the frame (blockIdx.y is the block row); else the list of block rows to
// compute (a rank that blooms only what its own rows need, rt_expose_rows_device)
__global__ __launch_bounds__(BLOOM_T * BLOOM_T) void bloom_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                                  int W, int H, const double* __restrict__ kern,
                                                                  const int* __restrict__ block_rows) {
    __shared__ double tile[BLOOM_TILE][BLOOM_TILE + 1];
    const int ch = blockIdx.z;
    const int by = block_rows ? block_rows[blockIdx.y] : (int)blockIdx.y;
    const int x0 = blockIdx.x * BLOOM_T - BLOOM_R, y0 = by * BLOOM_T - BLOOM_R;
    bool lit = false;
    for (int i = threadIdx.x; i < BLOOM_TILE * BLOOM_TILE; i += BLOOM_T * BLOOM_T) {
        const int ty = i / BLOOM_TILE, tx = i - ty * BLOOM_TILE;
        const int x = x0 + tx, y = y0 + ty;
        const double v = (x >= 0 && x < W && y >= 0 && y < H) ? src[((size_t)y * W + x) * 3 + ch] : 0.0;
        tile[ty][tx] = v;
        lit |= (v != 0.0);
    }
    // a tile without a lit source pixel (most of a frame: only emitters and highlights saturate) sums 1681 products
    // +0.0 * k into +0.0; the barrier doubles as that test
    const bool any_lit = __syncthreads_or(lit ? 1 : 0) != 0;
    const int lx = threadIdx.x % BLOOM_T, ly = threadIdx.x / BLOOM_T;
    double c = 0.0;
    if (any_lit) {
        for (int ky = 0; ky < BLOOM_K; ky++) {
            const double* krow = kern + ky * BLOOM_K;
#pragma unroll
            for (int kx = 0; kx < BLOOM_K; kx++) c = c + tile[ly + ky][lx + kx] * krow[kx];
        }
    }
    const int x = blockIdx.x * BLOOM_T + lx, y = by * BLOOM_T + ly;
    if (x < W && y < H) dst[((size_t)y * W + x) * 3 + ch] = c;
}

DEV int to_int_x86(double v);

// ---- expose() before the gather: every rank blooms and packs the rows it rendered -----------------------------
// B0 (main.cpp:198-205) is three bits per pixel.  Each rank reduces its rows to those bits (expose_bits_kernel), the
// ranks all-gather them (1 byte per pixel), and each rank then runs the blur passes on the 16-row blocks its own rows
// depend on: pass p of n needs pass p-1 within 20 rows, so the block set grows by two block rows per pass going
// backwards from the rank's own blocks (the halo).  Buffers are indexed by FRAME position, so a tap outside the frame is
// a tap outside the frame here too.  Every pixel goes through the same operations as in rt_expose_device.

// global row of a rank's packed row lr (cyclic row tiles)
DEV int tile_global_row(int lr, int tile_rows, int rank, int world) {
    return ((lr / tile_rows) * world + rank) * tile_rows + lr % tile_rows;
}

__global__ void expose_bits_kernel(const double* rgb, unsigned char* bits, size_t n, double exposure) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned b = 0;
        for (int k = 0; k < 3; k++) if (rgb[3 * i + k] * exposure > 1.0) b |= 1u << k;
        bits[i] = (unsigned char)b;
    }
}

// B0 of the listed block rows from the gathered bits: bits_all[r][lr][x] = rank r's packed row lr
__global__ void expose_b0_blocks_kernel(const unsigned char* __restrict__ bits_all, double* __restrict__ b0, int W, int H,
                                        int tile_rows, int world, int pad_rows, double exposure,
                                        const int* __restrict__ block_rows) {
    const int by = block_rows[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * BLOOM_T) return;
    const int y = by * BLOOM_T + t / W, x = t % W;
    if (y >= H) return;
    const int tile = y / tile_rows, r = tile % world, lr = (tile / world) * tile_rows + y % tile_rows;
    const unsigned b = bits_all[((size_t)r * pad_rows + lr) * W + x];
    double* o = b0 + ((size_t)y * W + x) * 3;
    for (int k = 0; k < 3; k++) o[k] = ((b >> k) & 1u) ? exposure : 0.0;
}

// expose_pack_kernel for a rank's packed rows; the bloom is read at the pixel's frame position
__global__ void expose_pack_rows_kernel(const double* rgb, const double* bloom, int* img, int W, int rows, int tile_rows,
                                        int rank, int world, double exposure) {
    const double mx = 1.0 / exposure;
    const size_t n = (size_t)rows * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lr = (int)(i / (size_t)W), x = (int)(i - (size_t)lr * W);
        const size_t f = (size_t)tile_global_row(lr, tile_rows, rank, world) * W + x;
        int v[3];
        for (int k = 0; k < 3; k++) {
            const double col = rgb[3 * i + k] + bloom[3 * f + k];
            int q = to_int_x86(255.0 * col / mx);
            if (q > 255) q = 255;
            v[k] = q;
        }
        img[i] = (int)(((unsigned)v[0] << 16) + ((unsigned)v[1] << 8) + (unsigned)v[2]);
    }
}

// (int)x as the reference's compiler converts it (cvttsd2si): truncation, 0x80000000 when out of range or NaN
DEV int to_int_x86(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000u;
    return (int)v;
}

// main.cpp:225-227 + 237-241 with Color::to_int (geometry.cpp:15-23).  bloom == nullptr: no blur pass was
// asked for and the reference adds B0 itself.
__global__ void expose_pack_kernel(const double* rgb, const double* bloom, int* img, size_t n, double exposure) {
    const double mx = 1.0 / exposure;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int v[3];
        for (int k = 0; k < 3; k++) {
            const double c = rgb[3 * i + k];
            const double b = bloom ? bloom[3 * i + k] : ((c * exposure > 1.0) ? exposure : 0.0);
            const double col = c + b;
            int q = to_int_x86(255.0 * col / mx);
            if (q > 255) q = 255;
            v[k] = q;
        }
        img[i] = (int)(((unsigned)v[0] << 16) + ((unsigned)v[1] << 8) + (unsigned)v[2]);
    }
}

// fn: 0 sin, 1 cos, 2 acos, 3 pow, 4 x/y, 5 sqrt(x), 6 next RNG state of (uint64)x bits
__global__ void selftest_math_kernel(int fn, int n, const double* x, const double* y, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (fn) {
        case 0: r = rtm::sind(x[i]); break;
        case 1: r = rtm::cosd(x[i]); break;
        case 2: r = rtm::acosd(x[i]); break;
        case 3: r = rtm::powd(x[i], y[i]); break;
        case 4: r = x[i] / y[i]; break;
        case 5: r = __builtin_sqrt(x[i]); break;
        default: { unsigned s = (unsigned)__double_as_longlong(x[i]); rng_next(s); r = __longlong_as_double((long long)s); } break;
    }
    out[i] = r;
}

// broad-phase self-test: n independent (ray, record) pairs; flag[i] = 1 if the float test keeps the pair
__global__ void selftest_broad_kernel(int n, const double* o, const double* d, const float4* rec, float cmax,
                                      double tx, double ty, double tz, int* flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double T[3] = {tx, ty, tz};
    const RayRec r = broad_setup(dmk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dmk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), T, cmax);
    float hb, nq;
    const float sv = broad_s(rec[i], r.a, r.b, hb, nq);
    int f = 0;
    if (sv >= r.b.w) f = broad_behind(hb, nq, r.c) ? 2 : 1;   // 2 = dropped as "behind"
    flag[i] = f;
}

}  // namespace

// ======================================================================= host side
constexpr int RT_MAX_SUB = 3;   // concurrent sub-pipelines (RT_OPT_OVERLAP); with the caller's stream they fit the 4 hardware queues of a process

struct rt_context {
    int device = 0;
    int cu_count = 0;
    std::string err;
    bool have_scene = false;
    // frame parameters
    int width = 0, height = 0, spp = 0, depth = 0, num_threads = 1;
    double scale = 0, lens = 0, focal = 0, exposure = 1;
    double cam[3] = {0, 0, 40};
    int num_blurs = 0;
    unsigned frame_seed = 0;
    int has_light = 0, light_draws = 0;
    bool regular_paths = false;   // every path ends within 2*depth+1 segments (an emitter exists, no index-1 glass)
    int n_prims = 0;
    uint64_t n_spheres = 0, n_planes = 0, n_faces = 0, n_face_verts = 0;
    // device scene
    void* d_records = nullptr;
    void* d_chunks = nullptr;
    void* d_prims = nullptr;
    void* d_verts = nullptr;
    void* d_mats = nullptr;
    void* d_always = nullptr; int n_always = 0, always_doubles = 0;
    // uniform grid over the bounded solids (walk_kernel); grid_ok: built and usable for this scene
    uint4* d_cells = nullptr; float4* d_item_rec = nullptr; uint2* d_item_aux = nullptr;
    GridD grid{}; bool grid_ok = false; size_t grid_items = 0;
    int nchunks = 0;
    int max_chunk_records = 0;
    int n_records_padded = 0;  // records incl. the NaN padding of every LDS window
    float cmax = 0;
    double T[3] = {0, 0, 0};
    unsigned* d_ctrl = nullptr;
    unsigned long long* d_stream_rays = nullptr; int stream_cap = 0;
    // pipeline state (per-pixel mode): one set per concurrent sub-pipeline, sized for `slots` pixel slots and `verts`
    // stack entries
    struct PipeBufs {
        size_t slots = 0; int verts = 0; int iters = 0;
        int* si[2] = {nullptr, nullptr}; double2* sd[2] = {nullptr, nullptr}; double* stack = nullptr; double* acc = nullptr;
        unsigned* queue[2] = {nullptr, nullptr};
        unsigned* counts = nullptr;
        float4* rec[2] = {nullptr, nullptr};
        uint2* cand = nullptr; unsigned* cand_cnt = nullptr;
    } pb[RT_MAX_SUB];
    hipStream_t sub_stream[RT_MAX_SUB] = {};   // overlap mode: the sub-pipelines' streams
    hipEvent_t fork_ev = nullptr, join_ev[RT_MAX_SUB] = {};
    std::vector<hipEvent_t> broad_ev;   // pairs around every broad_kernel launch when timing_detail is on
    int broad_ev_used = 0;
    // scratch outputs of rt_render
    double* d_rgb = nullptr; size_t d_rgb_bytes = 0;
    int32_t* d_hits = nullptr; size_t d_hits_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double* d_bloom[2] = {nullptr, nullptr}; size_t bloom_doubles = 0;   // rt_expose_device scratch
    double* d_bloom_kern = nullptr;
    // rt_expose_rows_device: block-row lists of the blur passes for one (height, tile_rows, rank, world, num_blurs)
    int* d_bloom_lists = nullptr;
    std::vector<int> bloom_list_ofs;             // pass p's list = d_bloom_lists[ofs[p] .. ofs[p+1])  (p = 0: B0)
    int bloom_list_key[5] = {-1, -1, -1, -1, -1};
    bool events_pending = false;
    int last_streams = 0;
    rt_timing timing{};
    // options
    int mode = RT_MODE_PIPELINE;
    int timing_detail = 0;
    int block = 256;           // single-kernel renderer: workgroup size
    int broad_block = 512;     // broad_kernel workgroup size
    int grid_mult = 0;         // workgroups per CU, 0 = ask the occupancy API
    int chunk_records = 8192;  // LDS window in records (16 bytes each)
    size_t max_slots = (size_t)1 << 25;
    int overlap = 2;           // pipeline: 2 = two half-frame sub-pipelines on two streams (broad of one beside shade of the other)
    int accel = 0;             // pipeline: 1 = candidate lists from the grid walk instead of the scan of every record
    double grid_density = 1.0; // cells per listed solid the grid aims at (RT_GRID_DENSITY); 0.5 to 8 measured: flat optimum, 1 best on C4
};

namespace {

int fail(rt_context* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define HIP_TRY(h, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(h, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

void free_scene(rt_context* h) {
    hipFree(h->d_records); hipFree(h->d_chunks); hipFree(h->d_prims); hipFree(h->d_verts); hipFree(h->d_mats); hipFree(h->d_always);
    hipFree(h->d_cells); hipFree(h->d_item_rec); hipFree(h->d_item_aux);
    h->d_cells = nullptr; h->d_item_rec = nullptr; h->d_item_aux = nullptr; h->grid_ok = false;
    h->d_records = h->d_chunks = h->d_prims = h->d_verts = h->d_mats = h->d_always = nullptr;
    h->have_scene = false;
}

void free_pipeline(rt_context* h) {
    for (auto& b : h->pb) {
        for (int q = 0; q < 2; q++) { hipFree(b.si[q]); hipFree(b.sd[q]); hipFree(b.queue[q]); hipFree(b.rec[q]); }
        hipFree(b.stack); hipFree(b.counts); hipFree(b.acc); hipFree(b.cand); hipFree(b.cand_cnt);
        b = rt_context::PipeBufs();
    }
}

// smallest float >= x
float round_up_f(double x) {
    float f = (float)x;
    if ((double)f < x) f = nextafterf(f, INFINITY);
    return f;
}

// the broad-phase record of a bounding sphere (centre c relative to T, radius R): (c, R^2 - |c|^2) with the
// radius inflated for the rounding of the centre to float; returns |c| + R for the scene-wide cmax
double make_record(const double* c, double R, float* out) {
    double c2 = 0, cn = 0;
    for (int k = 0; k < 3; k++) {
        out[k] = (float)c[k];
        c2 += (double)out[k] * out[k];
        cn += c[k] * c[k];
    }
    const double Rr = R * (1.0 + 1e-6) + 2.0 * sqrt(cn) * 5.9604645e-8 * 1.7320508;
    out[3] = round_up_f(Rr * Rr - c2);
    return sqrt(cn) + Rr;
}


// Uniform grid over the bounded solids that are not in the always table, for walk_kernel.  bc/br: bounding sphere of
// every solid (radius < 0: unbounded), T: the translation of the float coordinates.  A solid is listed in every cell
// that the box of its bounding sphere, grown by `pad`, overlaps.  Returns false when the scene cannot use the grid
// (an unbounded solid outside the always table would have to be listed everywhere).
struct GridHost {
    GridD g;
    size_t items = 0;                 // (solid, cell) pairs listed
    std::vector<unsigned> cells;      // 8 per cell (GridD::cells)
    std::vector<float> item_rec;      // 4 per further item
    std::vector<unsigned> item_aux;   // 2 per further item
};

bool build_grid(int n_prims, const std::vector<PrimD>& pd, const std::vector<double>& bc, const std::vector<double>& br,
                const std::vector<char>& always, const double* T, double density, GridHost& out) {
    const double U = 5.9604645e-8;
    std::vector<int> ids;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < n_prims; i++) {
        if (always[i]) continue;
        if (br[i] < 0) return false;
        ids.push_back(i);
        for (int k = 0; k < 3; k++) {
            const double c = bc[(size_t)i * 3 + k] - T[k];
            if (c - br[i] < lo[k]) lo[k] = c - br[i];
            if (c + br[i] > hi[k]) hi[k] = c + br[i];
        }
    }
    GridD& g = out.g;
    memset(&g, 0, sizeof(g));
    if (ids.empty()) {                     // one empty cell: every walk ends at once with an empty list
        for (int k = 0; k < 3; k++) { g.lo[k] = -1.0f; g.hi[k] = 1.0f; g.cell[k] = 2.0f; g.inv_cell[k] = 0.5f; g.n[k] = 1; }
        g.pad = 1e-3f; g.far = 1e30f;
        out.cells.assign(8, 0u);
        out.item_rec.assign(4, 0.0f); out.item_aux.assign(2, 0u);
        return true;
    }
    double reach = 0;
    for (int k = 0; k < 3; k++) { reach = fmax(reach, fabs(lo[k])); reach = fmax(reach, fabs(hi[k])); }
    if (!(reach < 1e15)) return false;
    if (reach < 1e-6) reach = 1e-6;
    // resolution: about `density` cells per listed solid, cells as cubic as the box allows
    double ext[3], vol = 1, emax = 0;
    for (int k = 0; k < 3; k++) { ext[k] = hi[k] - lo[k]; emax = fmax(emax, ext[k]); }
    for (int k = 0; k < 3; k++) { if (ext[k] < 1e-3 * emax) { const double grow = 0.5 * (1e-3 * emax - ext[k]); lo[k] -= grow; hi[k] += grow; ext[k] = hi[k] - lo[k]; } vol *= ext[k]; }
    const double per_len = cbrt(density * (double)ids.size() / vol);
    size_t cells = 1;
    double cell_min = 1e300;
    for (int k = 0; k < 3; k++) {
        int n = (int)ceil(ext[k] * per_len);
        n = n < 1 ? 1 : (n > 256 ? 256 : n);
        g.n[k] = n;
        cells *= (size_t)n;
        cell_min = fmin(cell_min, ext[k] / n);
    }
    // pad: what float rounding can move a cell boundary or a point of the ray, 128 u (|origin| + reach) with a wide
    // margin (the walk's own count is about 11 u).  Origins up to `far` from T are walked; `far` is as large as a
    // pad of 2 % of the smallest cell allows, at least 8 and at most 1024 times the grid's reach.
    double far = 0.02 * cell_min / (128.0 * U) - reach;
    far = fmin(fmax(far, 8.0 * reach), 1024.0 * reach);
    const double pad = 128.0 * U * (far + reach);
    for (int k = 0; k < 3; k++) { lo[k] -= 2.0 * pad; hi[k] += 2.0 * pad; }
    // the device recomputes a boundary as fma((float)i, cell, lo): lists are built from the same float values, and the
    // upper corner is the boundary of index n
    for (int k = 0; k < 3; k++) {
        g.lo[k] = (float)lo[k];
        if ((double)g.lo[k] > lo[k]) g.lo[k] = nextafterf(g.lo[k], -INFINITY);
        g.cell[k] = round_up_f((hi[k] - (double)g.lo[k]) / g.n[k]);
        g.inv_cell[k] = 1.0f / g.cell[k];
        g.hi[k] = fmaf((float)g.n[k], g.cell[k], g.lo[k]);
    }
    g.pad = round_up_f(pad); g.far = (float)far;
    auto cell_range = [&](int k, double a, double b, int& c0, int& c1) {
        c0 = (int)floor((a - (double)g.lo[k]) / (double)g.cell[k]);
        c1 = (int)floor((b - (double)g.lo[k]) / (double)g.cell[k]);
        if (c0 < 0) c0 = 0;
        if (c1 >= g.n[k]) c1 = g.n[k] - 1;
        if (c1 < c0) c1 = c0;
    };
    std::vector<unsigned> cnt(cells, 0u), more(cells, 0u);
    out.cells.assign(cells * 8, 0u);
    for (int pass = 0; pass < 2; pass++) {
        for (size_t q = 0; q < ids.size(); q++) {
            const int i = ids[q];
            int a0[3], a1[3];
            double cc[3];
            for (int k = 0; k < 3; k++) {
                cc[k] = bc[(size_t)i * 3 + k] - T[k];
                cell_range(k, cc[k] - br[i] - pad, cc[k] + br[i] + pad, a0[k], a1[k]);
            }
            float r4[4] = {0, 0, 0, 0};
            float r2 = -1.0f;
            if (pass == 1) {
                make_record(cc, br[i], r4);
                if (pd[i].kind == RT_KIND_SPHERE) {      // lower bound of r^2 for the certain-hit test
                    const double v = pd[i].radius * pd[i].radius * (1.0 - 1e-6);
                    r2 = (float)v;
                    if ((double)r2 > v) r2 = nextafterf(r2, -INFINITY);
                    if (!(r2 >= 0.0f)) r2 = 0.0f;
                }
            }
            for (int z = a0[2]; z <= a1[2]; z++) for (int y = a0[1]; y <= a1[1]; y++) for (int x = a0[0]; x <= a1[0]; x++) {
                const size_t ci = ((size_t)z * g.n[1] + y) * g.n[0] + x;
                if (pass == 0) { cnt[ci]++; continue; }
                unsigned* cr = &out.cells[ci * 8];
                const unsigned k = cr[6]++;
                if (k == 0) {
                    memcpy(cr, r4, 16);
                    cr[4] = (unsigned)i; memcpy(&cr[5], &r2, 4); cr[7] = more[ci];
                } else {
                    const size_t at = (size_t)more[ci] + k - 1;
                    memcpy(&out.item_rec[at * 4], r4, 16);
                    out.item_aux[at * 2] = (unsigned)i; memcpy(&out.item_aux[at * 2 + 1], &r2, 4);
                }
            }
        }
        if (pass == 0) {
            size_t extra = 0, total = 0;
            for (size_t c = 0; c < cells; c++) { more[c] = (unsigned)extra; if (cnt[c] > 1) extra += cnt[c] - 1; total += cnt[c]; }
            if (total > (1u << 28)) return false;
            out.items = total;
            out.item_rec.assign(extra * 4 + 4, 0.0f);
            out.item_aux.assign(extra * 2 + 2, 0u);
        }
    }
    return true;
}

void fill_kparams(rt_context* h, KParams& kp) {
    memset(&kp, 0, sizeof(kp));
    kp.width = h->width; kp.height = h->height; kp.spp = h->spp; kp.depth = h->depth;
    kp.scale = h->scale; kp.lens = h->lens; kp.focal = h->focal; kp.frame_seed = h->frame_seed;
    for (int k = 0; k < 3; k++) kp.cam[k] = h->cam[k];
    kp.has_light = h->has_light; kp.light_draws = h->light_draws;
    kp.tile_rows = 1; kp.rank = 0; kp.world = 1;
    kp.nchunks = h->nchunks;
    kp.n_prims = h->n_prims;
    kp.max_verts = RT_MAXV;
    kp.cmax = h->cmax;
    for (int k = 0; k < 3; k++) kp.T[k] = h->T[k];
    kp.records = (const float4*)h->d_records;
    kp.chunks = (const ChunkDesc*)h->d_chunks;
    kp.prims = (const PrimD*)h->d_prims;
    kp.verts = (const FaceV*)h->d_verts;
    kp.mats = (const MatD*)h->d_mats;
    kp.always = (const double*)h->d_always; kp.n_always = h->n_always; kp.always_doubles = h->always_doubles;
    kp.ctrl = h->d_ctrl;
}

template <int BLOCK>
hipError_t launch_render(rt_context* h, const KParams& kp, hipStream_t st) {
    auto kern = render_kernel<BLOCK>;
    size_t lds = (size_t)h->max_chunk_records * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = h->grid_mult;
    if (per_cu <= 0) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
    }
    unsigned grid = (unsigned)(h->cu_count * per_cu);
    unsigned need = (kp.total_items + BLOCK - 1) / BLOCK;
    if (need < grid) grid = need ? need : 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, st, kp);
    return hipGetLastError();
}

template <int BLOCK>
hipError_t launch_broad(rt_context* h, const WfParams& w, unsigned grid, hipStream_t st) {
    size_t lds = (size_t)h->max_chunk_records * sizeof(float4);
    hipLaunchKernelGGL(broad_kernel<BLOCK>, dim3(grid), dim3(BLOCK), lds, st, w);
    return hipGetLastError();
}

template <int BLOCK>
hipError_t broad_grid(rt_context* h, unsigned* grid) {
    auto kern = broad_kernel<BLOCK>;
    size_t lds = (size_t)h->max_chunk_records * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = h->grid_mult;
    if (per_cu <= 0) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
    }
    *grid = (unsigned)(h->cu_count * per_cu);
    return hipSuccess;
}

}  // namespace

extern "C" {

int rt_create(rt_handle* out, int device) {
    if (!out) return RT_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return RT_ERR_NO_DEVICE;
    rt_context* h = new rt_context();
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete h; return RT_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete h; return RT_ERR_NO_DEVICE; }
    h->cu_count = prop.multiProcessorCount;
    if (hipMalloc((void**)&h->d_ctrl, 256) != hipSuccess) { delete h; return RT_ERR_HIP; }
    hipEventCreate(&h->ev0);
    hipEventCreate(&h->ev1);
    // tuning knobs for experiments; rt_set_option is the API for the same values
    if (const char* e = getenv("RT_MODE")) rt_set_option(h, RT_OPT_MODE, strcmp(e, "mega") == 0 ? RT_MODE_SINGLE_KERNEL : RT_MODE_PIPELINE);
    if (const char* e = getenv("RT_BLOCK")) rt_set_option(h, RT_OPT_BLOCK, atoi(e));
    if (const char* e = getenv("RT_BROAD_BLOCK")) rt_set_option(h, RT_OPT_BROAD_BLOCK, atoi(e));
    if (const char* e = getenv("RT_WG_PER_CU")) rt_set_option(h, RT_OPT_WG_PER_CU, atoi(e));
    if (const char* e = getenv("RT_CHUNK_RECORDS")) rt_set_option(h, RT_OPT_CHUNK_RECORDS, atoi(e));
    if (const char* e = getenv("RT_OVERLAP")) rt_set_option(h, RT_OPT_OVERLAP, atoi(e));
    if (const char* e = getenv("RT_ACCEL")) rt_set_option(h, RT_OPT_ACCEL, atoi(e));
    if (const char* e = getenv("RT_GRID_DENSITY")) { const double v = atof(e); if (v > 0.01 && v < 100.0) h->grid_density = v; }
    *out = h;
    return RT_OK;
}

int rt_set_option(rt_handle h, int key, int value) {
    if (!h) return RT_ERR_ARG;
    switch (key) {
        case RT_OPT_MODE:
            if (value != RT_MODE_PIPELINE && value != RT_MODE_SINGLE_KERNEL) return fail(h, RT_ERR_ARG, "rt_set_option: bad mode");
            h->mode = value; return RT_OK;
        case RT_OPT_TIMING_DETAIL: h->timing_detail = value != 0; return RT_OK;
        case RT_OPT_FUSE_NARROW: return RT_OK;   // retired: the narrow phase always runs inside shade_kernel
        case RT_OPT_BLOCK:
            if (value != 256 && value != 512 && value != 1024) return fail(h, RT_ERR_ARG, "rt_set_option: block must be 256, 512 or 1024");
            h->block = value; return RT_OK;
        case RT_OPT_BROAD_BLOCK:
            if (value != 256 && value != 512 && value != 1024) return fail(h, RT_ERR_ARG, "rt_set_option: block must be 256, 512 or 1024");
            h->broad_block = value; return RT_OK;
        case RT_OPT_WG_PER_CU: h->grid_mult = value > 0 ? value : 0; return RT_OK;
        case RT_OPT_CHUNK_RECORDS:   // takes effect at the next rt_upload_scene
            if (value < 64 || value > 9728) return fail(h, RT_ERR_ARG, "rt_set_option: LDS window must be 64..9728 records");
            h->chunk_records = value / BROAD_GROUP * BROAD_GROUP; return RT_OK;
        case RT_OPT_MAX_SLOTS:
            if (value < 1) return fail(h, RT_ERR_ARG, "rt_set_option: max slots must be positive");
            h->max_slots = (size_t)value; return RT_OK;
        case RT_OPT_OVERLAP:
            if (value < 1 || value > RT_MAX_SUB) return fail(h, RT_ERR_ARG, "rt_set_option: overlap must be 1..3");
            h->overlap = value; return RT_OK;
        case RT_OPT_ACCEL:
            if (value != 0 && value != 1) return fail(h, RT_ERR_ARG, "rt_set_option: accel must be 0 or 1");
            h->accel = value; return RT_OK;
        default: return fail(h, RT_ERR_ARG, "rt_set_option: unknown key");
    }
}

void rt_destroy(rt_handle h) {
    if (!h) return;
    hipSetDevice(h->device);
    free_scene(h);
    free_pipeline(h);
    hipFree(h->d_bloom[0]); hipFree(h->d_bloom[1]); hipFree(h->d_bloom_kern); hipFree(h->d_bloom_lists);
    hipFree(h->d_ctrl); hipFree(h->d_rgb); hipFree(h->d_hits); hipFree(h->d_stream_rays);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->broad_ev) hipEventDestroy(e);
    for (int p = 0; p < RT_MAX_SUB; p++) { if (h->sub_stream[p]) hipStreamDestroy(h->sub_stream[p]); if (h->join_ev[p]) hipEventDestroy(h->join_ev[p]); }
    if (h->fork_ev) hipEventDestroy(h->fork_ev);
    delete h;
}

const char* rt_last_error(rt_handle h) { return h ? h->err.c_str() : "null handle"; }

int rt_rows_for_rank(int height, int tile_rows, int rank, int world) {
    if (height < 0 || tile_rows < 1 || world < 1 || rank < 0 || rank >= world) return RT_ERR_ARG;
    int rows = 0;
    for (long k = rank; k * tile_rows < height; k += world) {
        long r0 = k * tile_rows, r1 = r0 + tile_rows;
        if (r1 > height) r1 = height;
        rows += (int)(r1 - r0);
    }
    return rows;
}

int rt_upload_scene(rt_handle h, const double* params, int n_prims, const double* prims, int n_mats,
                    const double* mats, int n_verts, const double* verts) {
    if (!h) return RT_ERR_ARG;
    if (!params || n_prims < 0 || n_mats < 0 || n_verts < 0 || (n_prims && !prims) || (n_mats && !mats) || (n_verts && !verts))
        return fail(h, RT_ERR_ARG, "rt_upload_scene: null table or negative count");
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    hipEventRecord(t0, 0);
    const int W = (int)params[RT_PA_WIDTH], H = (int)params[RT_PA_HEIGHT];
    const int spp = (int)params[RT_PA_SPP], depth = (int)params[RT_PA_DEPTH];
    if (W < 1 || H < 1 || spp < 1 || depth < 0) return fail(h, RT_ERR_ARG, "rt_upload_scene: width, height, paths_per_pixel must be >= 1, depth >= 0");
    if (depth > RT_MAX_DEPTH) return fail(h, RT_ERR_SCENE, "rt_upload_scene: max_ray_depth exceeds RT_MAX_DEPTH");
    if ((uint64_t)W * (uint64_t)H >= (1ull << 31)) return fail(h, RT_ERR_SCENE, "rt_upload_scene: more than 2^31 pixels");

    // ---- materials
    std::vector<MatD> md((size_t)n_mats > 0 ? n_mats : 1);
    memset(md.data(), 0, md.size() * sizeof(MatD));
    for (int i = 0; i < n_mats; i++) {
        const double* m = mats + (size_t)i * RT_MT_STRIDE;
        MatD& o = md[i];
        for (int k = 0; k < 3; k++) {
            o.ref[k] = m[RT_MT_REF + k]; o.diff[k] = m[RT_MT_DIFF + k]; o.spec[k] = m[RT_MT_SPEC + k];
            o.rdiff[k] = m[RT_MT_RDIFF + k]; o.rspec[k] = m[RT_MT_RSPEC + k];
        }
        o.is_light = m[RT_MT_LIGHT] != 0.0;
        // generators() < gloss*generators.max() (main.cpp:48) for an integer draw g: g < ceil(gloss*max);
        // a NaN gloss never compares true.
        double thr = m[RT_MT_GLOSS] * 2147483646.0;
        unsigned it = 0;
        if (thr > 0.0) {
            if (thr >= 4294967295.0) it = 4294967295u;
            else { it = (unsigned)thr; if ((double)it < thr) it++; }
        }
        o.gloss_thr = it;
        o.sexp = m[RT_MT_SEXP];
        o.rindex = m[RT_MT_RINDEX];
    }

    // ---- exact tables + bounding spheres (centre, radius; radius < 0 = unbounded)
    std::vector<PrimD> pd((size_t)n_prims > 0 ? n_prims : 1);
    memset(pd.data(), 0, pd.size() * sizeof(PrimD));
    std::vector<double> bc((size_t)n_prims * 3 + 3, 0.0), br((size_t)n_prims + 1, -1.0);
    h->n_spheres = h->n_planes = h->n_faces = h->n_face_verts = 0;
    int first_light = -1;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < n_prims; i++) {
        const double* p = prims + (size_t)i * RT_PR_STRIDE;
        const int kind = (int)p[RT_PR_KIND];
        if (kind < 0 || kind > 2) return fail(h, RT_ERR_ARG, "rt_upload_scene: bad primitive kind");
        const int m0 = (int)p[RT_PR_MAT0], m1 = (int)p[RT_PR_MAT1];
        if (m0 < 0 || m0 >= n_mats || m1 < 0 || m1 >= n_mats) return fail(h, RT_ERR_ARG, "rt_upload_scene: material index out of range");
        PrimD& q = pd[i];
        q.kind = kind; q.mat0 = m0; q.mat1 = m1;
        for (int k = 0; k < 3; k++) { q.a[k] = p[RT_PR_A + k]; q.n[k] = p[RT_PR_N + k]; }
        q.radius = kind == RT_KIND_SPHERE ? p[RT_PR_RADIUS] : (double)NAN;   // see narrow_point
        if (p[RT_PR_LIGHT] != 0.0 && first_light < 0) first_light = i;
        bool bounded = false;
        if (kind == RT_KIND_SPHERE) {
            h->n_spheres++;
            if (std::isfinite(q.a[0]) && std::isfinite(q.a[1]) && std::isfinite(q.a[2]) && std::isfinite(q.radius)) {
                for (int k = 0; k < 3; k++) bc[(size_t)i * 3 + k] = q.a[k];
                br[i] = fabs(q.radius);
                bounded = true;
            }
        } else if (kind == RT_KIND_PLANE) {
            h->n_planes++;
        } else {
            const int nv = (int)p[RT_PR_VCNT], off = (int)p[RT_PR_VOFF];
            if (nv < 1 || nv > RT_MAX_FACE_VERTS || off < 0 || off + nv > n_verts)
                return fail(h, RT_ERR_SCENE, "rt_upload_scene: face vertex range invalid or more than RT_MAX_FACE_VERTS vertices");
            q.voff = off; q.vcnt = nv;
            h->n_faces++;
            h->n_face_verts += (uint64_t)nv;
            // The reference accepts a point of the plane (verts[0], normal) iff, for every i, it is not on the
            // other side of edge (v_i, v_i+1) than v_i+2.  For a planar convex polygon whose plane is the one
            // the normal describes that region is the polygon, which lies inside the sphere around its
            // vertices.  Anything else keeps an unbounded sphere (always a candidate).
            const double* v = verts + (size_t)off * 3;
            bool ok = nv >= 3;
            double c[3] = {0, 0, 0};
            for (int k = 0; k < nv && ok; k++) for (int a = 0; a < 3; a++) { if (!std::isfinite(v[3 * k + a])) ok = false; c[a] += v[3 * k + a] / nv; }
            double R = 0, ext = 0;
            for (int k = 0; k < nv && ok; k++) {
                double dd = 0;
                for (int a = 0; a < 3; a++) dd += (v[3 * k + a] - c[a]) * (v[3 * k + a] - c[a]);
                if (sqrt(dd) > R) R = sqrt(dd);
            }
            ext = R > 0 ? R : 1.0;
            for (int k = 0; k < nv && ok; k++) {          // planarity w.r.t. the given normal
                double dist = 0;
                for (int a = 0; a < 3; a++) dist += (v[3 * k + a] - v[a]) * q.n[a];
                if (!(fabs(dist) <= 1e-9 * ext)) ok = false;
            }
            for (int k = 0; k < nv && ok; k++) {          // convexity: all vertices on one side of every edge
                const double* a0 = v + 3 * k; const double* a1 = v + 3 * ((k + 1) % nv);
                double e[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]};
                double s[3] = {e[1] * q.n[2] - e[2] * q.n[1], e[2] * q.n[0] - e[0] * q.n[2], e[0] * q.n[1] - e[1] * q.n[0]};
                int sign = 0;
                for (int j = 0; j < nv && ok; j++) {
                    const double* b = v + 3 * j;
                    double t = (b[0] - a0[0]) * s[0] + (b[1] - a0[1]) * s[1] + (b[2] - a0[2]) * s[2];
                    if (fabs(t) <= 1e-9 * ext * ext) continue;
                    int sg = t > 0 ? 1 : -1;
                    if (sign == 0) sign = sg; else if (sg != sign) ok = false;
                }
            }
            if (ok) {
                for (int k = 0; k < 3; k++) bc[(size_t)i * 3 + k] = c[k];
                br[i] = R * (1.0 + 1e-6) + 1e-6 * ext;
                bounded = true;
            }
        }
        if (bounded) for (int k = 0; k < 3; k++) {
            double a = bc[(size_t)i * 3 + k];
            if (a - br[i] < lo[k]) lo[k] = a - br[i];
            if (a + br[i] > hi[k]) hi[k] = a + br[i];
        }
    }
    double T[3] = {0, 0, 0};
    if (lo[0] <= hi[0]) for (int k = 0; k < 3; k++) T[k] = 0.5 * (lo[k] + hi[k]);
    double cmax = 0;

    // ---- "always" solids: unbounded ones and those whose bounding sphere is a sizeable part of the scene (room walls,
    // floor, ceiling).  Nearly every ray would keep them, so they skip the broad phase: the narrow phase tests them for
    // every ray from a small table (LDS), and their records become padding.  Any choice is correct; this one is for speed.
    std::vector<char> always((size_t)n_prims + 1, 0);
    {
        double diag = 0;
        if (lo[0] <= hi[0]) for (int k = 0; k < 3; k++) diag += (hi[k] - lo[k]) * (hi[k] - lo[k]);
        const double big = 0.125 * sqrt(diag);      // radius >= a quarter of the scene's half-diagonal
        int cnt = 0, doubles = 0;
        for (int i = 0; i < n_prims; i++) {
            if (!(br[i] < 0 || br[i] >= big)) continue;
            const int need = ALW_HEAD + 8 * (pd[i].kind == RT_KIND_FACE ? pd[i].vcnt : 0);
            if (cnt >= 64 || doubles + need > 4096) continue;   // over budget: stays an ordinary record
            always[i] = 1; cnt++; doubles += need;
        }
    }

    // ---- broad-phase records in file order, chunked to the LDS window, padded with NaN records
    std::vector<float> rec;
    std::vector<ChunkDesc> chunks;
    const int cap = h->chunk_records;
    for (int i = 0; i < n_prims; i++) {
        if (i % cap == 0) {
            ChunkDesc c; c.gofs = (int)(rec.size() / 4); c.count = 0; c.first_index = i; c.pad = 0;
            chunks.push_back(c);
        }
        float r4[4];
        if (always[i]) {
            r4[0] = r4[1] = r4[2] = 0.0f; r4[3] = -INFINITY;       // tested for every ray by the narrow phase: never kept here
        } else if (br[i] < 0) {
            r4[0] = r4[1] = r4[2] = 0.0f; r4[3] = INFINITY;        // unbounded: always kept, never "behind"
        } else {
            double c[3];
            for (int k = 0; k < 3; k++) c[k] = bc[(size_t)i * 3 + k] - T[k];
            const double ext = make_record(c, br[i], r4);
            if (ext > cmax) cmax = ext;
        }
        rec.insert(rec.end(), r4, r4 + 4);
        chunks.back().count++;
    }
    // pad every LDS window to whole mask words inside the record stream
    {
        std::vector<float> padded;
        int src = 0;
        for (auto& c : chunks) {
            c.gofs = (int)(padded.size() / 4);
            padded.insert(padded.end(), rec.begin() + (size_t)src * 4, rec.begin() + (size_t)(src + c.count) * 4);
            src += c.count;
            // padding records (0, 0, 0, -inf): s = hb^2 - inf = -inf for every finite ray, never kept
            while (c.count % BROAD_GROUP) { padded.push_back(0.0f); padded.push_back(0.0f); padded.push_back(0.0f); padded.push_back(-INFINITY); c.count++; }
        }
        rec.swap(padded);
    }
    if (chunks.empty()) {   // empty scene: one window of padding keeps every loop and mask word well defined
        ChunkDesc c; c.gofs = 0; c.count = BROAD_GROUP; c.first_index = 0; c.pad = 0;
        chunks.push_back(c);
        rec.clear();
        for (int k = 0; k < BROAD_GROUP; k++) { rec.push_back(0.0f); rec.push_back(0.0f); rec.push_back(0.0f); rec.push_back(-INFINITY); }
    }
    int maxrec = BROAD_GROUP, totrec = 0;
    for (auto& c : chunks) { if (c.count > maxrec) maxrec = c.count; totrec = c.first_index + c.count; }
    h->n_records_padded = totrec;
    // face vertex table: vertex, edge to the next vertex and to_next, per face, in the reference's operation order
    // (geometry.cpp:163-164: (v_i - v_i+1).dot(v_i - v_i+2), Vector::dot accumulating from 0, left to right)
    std::vector<FaceV> vd((size_t)(n_verts > 0 ? n_verts : 1));
    memset(vd.data(), 0, vd.size() * sizeof(FaceV));
    for (int i = 0; i < n_prims; i++) {
        if (pd[i].kind != RT_KIND_FACE) continue;
        const int nv = pd[i].vcnt, off = pd[i].voff;
        for (int k = 0; k < nv; k++) {
            const double* vi = verts + (size_t)(off + k) * 3;
            const double* v1 = verts + (size_t)(off + (k + 1) % nv) * 3;
            const double* v2 = verts + (size_t)(off + (k + 2) % nv) * 3;
            FaceV& f = vd[(size_t)off + k];
            double t = 0.0;
            for (int a = 0; a < 3; a++) {
                f.v[a] = vi[a];
                f.e[a] = vi[a] - v1[a];
                const double w = vi[a] - v2[a];
                t = t + f.e[a] * w;
            }
            f.to_next = t;
        }
    }

    // the table of the "always" solids
    std::vector<double> alw;
    int n_always = 0;
    for (int i = 0; i < n_prims; i++) {
        if (!always[i]) continue;
        const int nv = pd[i].kind == RT_KIND_FACE ? pd[i].vcnt : 0;
        const double head[ALW_HEAD] = {(double)pd[i].kind, (double)i, (double)nv, 0.0, pd[i].a[0], pd[i].a[1], pd[i].a[2],
                                       pd[i].radius, pd[i].n[0], pd[i].n[1], pd[i].n[2], 0.0};
        alw.insert(alw.end(), head, head + ALW_HEAD);
        for (int k = 0; k < nv; k++) {
            const FaceV& f = vd[(size_t)pd[i].voff + k];
            const double v8[8] = {f.v[0], f.v[1], f.v[2], f.e[0], f.e[1], f.e[2], f.to_next, 0.0};
            alw.insert(alw.end(), v8, v8 + 8);
        }
        n_always++;
    }
    const int always_doubles = (int)alw.size();
    if (alw.empty()) alw.push_back(0.0);

    GridHost gh;
    const bool grid_ok = build_grid(n_prims, pd, bc, br, always, T, h->grid_density, gh);

    free_scene(h);
    if (grid_ok) {
        HIP_TRY(h, hipMalloc((void**)&h->d_cells, gh.cells.size() * sizeof(unsigned)));
        HIP_TRY(h, hipMalloc((void**)&h->d_item_rec, gh.item_rec.size() * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->d_item_aux, gh.item_aux.size() * sizeof(unsigned)));
        HIP_TRY(h, hipMemcpy(h->d_cells, gh.cells.data(), gh.cells.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_item_rec, gh.item_rec.data(), gh.item_rec.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_item_aux, gh.item_aux.data(), gh.item_aux.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        h->grid = gh.g;
        h->grid.cells = h->d_cells; h->grid.item_rec = h->d_item_rec; h->grid.item_aux = h->d_item_aux;
        h->grid_items = gh.items;
        h->grid_ok = true;
    }
    HIP_TRY(h, hipMalloc(&h->d_always, alw.size() * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->d_always, alw.data(), alw.size() * sizeof(double), hipMemcpyHostToDevice));
    h->n_always = n_always; h->always_doubles = always_doubles;
    HIP_TRY(h, hipMalloc(&h->d_records, rec.size() * sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->d_chunks, chunks.size() * sizeof(ChunkDesc)));
    HIP_TRY(h, hipMalloc(&h->d_prims, pd.size() * sizeof(PrimD)));
    HIP_TRY(h, hipMalloc(&h->d_verts, vd.size() * sizeof(FaceV)));
    HIP_TRY(h, hipMalloc(&h->d_mats, md.size() * sizeof(MatD)));
    HIP_TRY(h, hipMemcpy(h->d_records, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_chunks, chunks.data(), chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_prims, pd.data(), pd.size() * sizeof(PrimD), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_verts, vd.data(), vd.size() * sizeof(FaceV), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_mats, md.data(), md.size() * sizeof(MatD), hipMemcpyHostToDevice));

    h->width = W; h->height = H; h->spp = spp; h->depth = depth;
    h->num_threads = (int)params[RT_PA_THREADS] > 0 ? (int)params[RT_PA_THREADS] : 1;
    h->scale = params[RT_PA_SCALE]; h->lens = params[RT_PA_LENS]; h->focal = params[RT_PA_FOCAL];
    for (int k = 0; k < 3; k++) h->cam[k] = params[RT_PA_CAM_X + k];
    h->exposure = params[RT_PA_EXPOSURE]; h->num_blurs = (int)params[RT_PA_BLURS] > 0 ? (int)params[RT_PA_BLURS] : 0;
    h->frame_seed = (unsigned)params[RT_PA_SEED];
    h->has_light = first_light >= 0;
    h->light_draws = 0;
    if (first_light >= 0) {
        int k = (int)prims[(size_t)first_light * RT_PR_STRIDE + RT_PR_KIND];
        h->light_draws = (k == RT_KIND_PLANE) ? 0 : 2;   // rand_point, geometry.cpp:120,145-150,175-179
    }
    // paths are bounded by 2*depth+1 segments when an emitter exists (main.cpp:137-153 ends a path at depth 0)
    // and no transparent material has an index within 1e-4 of 1 (such a hit would always count as "leaving")
    bool regular = first_light >= 0;
    for (int i = 0; i < n_mats; i++) if (fabs(mats[(size_t)i * RT_MT_STRIDE + RT_MT_RINDEX] - 1.0) < 0.0002) regular = false;
    h->regular_paths = regular;
    h->n_prims = n_prims;
    h->nchunks = (int)chunks.size();
    h->max_chunk_records = maxrec;
    h->cmax = (float)(cmax * (1.0 + 1e-6)) ;
    for (int k = 0; k < 3; k++) h->T[k] = T[k];
    h->have_scene = true;
    hipEventRecord(t1, 0);
    hipEventSynchronize(t1);
    float ms = 0;
    hipEventElapsedTime(&ms, t0, t1);
    h->timing.upload_ms = ms;
    hipEventDestroy(t0); hipEventDestroy(t1);
    return RT_OK;
}

static int render_single_kernel(rt_handle h, int stream_mode, int tile_rows, int rank, int world, double* d_rgb,
                                int32_t* d_hit_ids, hipStream_t st) {
    KParams kp;
    fill_kparams(h, kp);
    kp.tile_rows = tile_rows; kp.rank = rank; kp.world = world;
    kp.stream_mode = stream_mode;
    if (stream_mode) {
        kp.n_streams = h->num_threads;
        kp.total_items = (unsigned)h->num_threads;
        if (h->stream_cap < h->num_threads) {
            hipFree(h->d_stream_rays); h->d_stream_rays = nullptr; h->stream_cap = 0;
            HIP_TRY(h, hipMalloc((void**)&h->d_stream_rays, (size_t)h->num_threads * 8));
            h->stream_cap = h->num_threads;
        }
        HIP_TRY(h, hipMemsetAsync(h->d_stream_rays, 0, (size_t)h->num_threads * 8, st));
        kp.stream_rays = h->d_stream_rays;
    } else {
        const int rows = rt_rows_for_rank(h->height, tile_rows, rank, world);
        kp.total_items = (unsigned)((uint64_t)rows * (uint64_t)h->width);
    }
    h->last_streams = stream_mode ? h->num_threads : 0;
    kp.rgb = d_rgb; kp.hit_ids = d_hit_ids;
    HIP_TRY(h, hipMemsetAsync(h->d_ctrl, 0, 256, st));
    HIP_TRY(h, hipEventRecord(h->ev0, st));
    if (kp.total_items > 0) {
        hipError_t e;
        if (h->block == 1024) e = launch_render<1024>(h, kp, st);
        else if (h->block == 512) e = launch_render<512>(h, kp, st);
        else e = launch_render<256>(h, kp, st);
        if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("render kernel launch: ") + hipGetErrorString(e));
    }
    HIP_TRY(h, hipEventRecord(h->ev1, st));
    h->broad_ev_used = 0;
    h->events_pending = true;
    return RT_OK;
}

// (re)allocate one sub-pipeline's per-slot state for `slots` pixel slots
static int ensure_pipeline(rt_handle h, rt_context::PipeBufs& b, size_t slots, int verts, int iters) {
    if (b.slots >= slots && b.verts >= verts && b.iters >= iters) return RT_OK;
    {
        for (int q = 0; q < 2; q++) { hipFree(b.si[q]); hipFree(b.sd[q]); hipFree(b.queue[q]); hipFree(b.rec[q]); }
        hipFree(b.stack); hipFree(b.counts); hipFree(b.acc); hipFree(b.cand); hipFree(b.cand_cnt);
        b = rt_context::PipeBufs();
    }
    for (int q = 0; q < 2; q++) HIP_TRY(h, hipMalloc((void**)&b.si[q], slots * SI_COUNT * sizeof(int)));
    for (int q = 0; q < 2; q++) HIP_TRY(h, hipMalloc((void**)&b.sd[q], slots * SD_COUNT * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&b.stack, slots * (size_t)verts * 4 * sizeof(double)));
    for (int q = 0; q < 2; q++) HIP_TRY(h, hipMalloc((void**)&b.queue[q], slots * sizeof(unsigned)));
    HIP_TRY(h, hipMalloc((void**)&b.counts, ((size_t)iters + 2) * sizeof(unsigned)));
    for (int q = 0; q < 2; q++) HIP_TRY(h, hipMalloc((void**)&b.rec[q], slots * sizeof(float4)));
    HIP_TRY(h, hipMalloc((void**)&b.acc, slots * 3 * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&b.cand, slots * WF_CAND_CAP * sizeof(uint2)));
    HIP_TRY(h, hipMalloc((void**)&b.cand_cnt, 2 * slots * sizeof(unsigned)));
    b.slots = slots; b.verts = verts; b.iters = iters;
    return RT_OK;
}

__global__ void set_count_kernel(unsigned* counts, unsigned n_counts, unsigned first) {
    for (unsigned i = threadIdx.x; i < n_counts; i += blockDim.x) counts[i] = (i == 0) ? first : 0u;
}

static int render_pipeline(rt_handle h, int tile_rows, int rank, int world, double* d_rgb, int32_t* d_hit_ids, hipStream_t st) {
    const int rows = rt_rows_for_rank(h->height, tile_rows, rank, world);
    const uint64_t total = (uint64_t)rows * (uint64_t)h->width;
    // longest possible path in segments, and with it the number of broad/shade rounds a pixel can need
    const int verts = h->regular_paths ? (2 * h->depth > 0 ? 2 * h->depth : 1) : RT_MAXV;
    const int iters = h->spp * (verts + 1);
    // Sub-pipelines (two by default), each with an equal share of the rank's pixels and its own stream: a kernel of one fills the CUs that
    // the tail of the other's kernel leaves idle (the late, short rounds above all).  Measured on the 4K frame: 8 %
    // less frame time.  Forcing the two kernel types side by side on every CU (one shade workgroup + broad workgroups,
    // sized through LDS requests) or alternating the broad kernels by events was slower than letting the dispatcher mix.
    // With per-launch timing on, the kernels run one after the other (a launch's measured duration is then its own).
    const int nsub = (h->overlap >= 2 && !h->timing_detail && total >= (1u << 16)) ? h->overlap : 1;
    uint64_t sub_begin[RT_MAX_SUB + 1];
    for (int p = 0; p <= nsub; p++) sub_begin[p] = p == nsub ? total : (total * p / nsub + 63) / 64 * 64;
    size_t slots_cap[RT_MAX_SUB];
    WfParams wp[RT_MAX_SUB];
    for (int p = 0; p < nsub; p++) {
        const uint64_t cnt = sub_begin[p + 1] - sub_begin[p];
        const size_t cap = h->max_slots / (size_t)nsub ? h->max_slots / (size_t)nsub : 1;
        slots_cap[p] = cnt < cap ? (size_t)(cnt ? cnt : 1) : cap;
        int rc = ensure_pipeline(h, h->pb[p], slots_cap[p], verts, iters);
        if (rc != RT_OK) return rc;
        WfParams& w = wp[p];
        const rt_context::PipeBufs& b = h->pb[p];
        fill_kparams(h, w.kp);
        w.kp.tile_rows = tile_rows; w.kp.rank = rank; w.kp.world = world;
        w.kp.max_verts = verts;
        w.kp.total_items = (unsigned)total;
        w.kp.rgb = d_rgb; w.kp.hit_ids = d_hit_ids;
        for (int q = 0; q < 2; q++) { w.si[q] = b.si[q]; w.sd[q] = b.sd[q]; w.queue[q] = b.queue[q]; w.rec[q] = b.rec[q]; }
        w.stack = b.stack; w.counts = b.counts; w.acc = b.acc;
        w.cand = b.cand; w.cand_cnt = b.cand_cnt;
    }
    unsigned bgrid = 1;
    hipError_t e;
    if (h->broad_block == 1024) e = broad_grid<1024>(h, &bgrid);
    else if (h->broad_block == 256) e = broad_grid<256>(h, &bgrid);
    else e = broad_grid<512>(h, &bgrid);
    if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("broad kernel setup: ") + hipGetErrorString(e));
    const bool use_grid = h->accel && h->grid_ok;
    unsigned sgrid = (unsigned)h->cu_count * 16u;
    if (const char* e = getenv("RT_SHADE_GRID")) sgrid = (unsigned)h->cu_count * (unsigned)atoi(e);
    h->last_streams = 0;
    h->broad_ev_used = 0;
    if (h->timing_detail) {
        const size_t nb = (total + slots_cap[0] - 1) / slots_cap[0];
        const size_t need = (size_t)2 * iters * (nb ? nb : 1);
        while (h->broad_ev.size() < need) { hipEvent_t ev; HIP_TRY(h, hipEventCreate(&ev)); h->broad_ev.push_back(ev); }
    }
    HIP_TRY(h, hipMemsetAsync(h->d_ctrl, 0, 256, st));
    HIP_TRY(h, hipEventRecord(h->ev0, st));
    hipStream_t ss[RT_MAX_SUB] = {st, st, st};
    if (nsub > 1) {
        for (int p = 0; p < nsub; p++) {
            if (!h->sub_stream[p]) HIP_TRY(h, hipStreamCreateWithFlags(&h->sub_stream[p], hipStreamNonBlocking));
            if (!h->join_ev[p]) HIP_TRY(h, hipEventCreateWithFlags(&h->join_ev[p], hipEventDisableTiming));
            ss[p] = h->sub_stream[p];
        }
        if (!h->fork_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->fork_ev, st));
        for (int p = 0; p < nsub; p++) HIP_TRY(h, hipStreamWaitEvent(ss[p], h->fork_ev, 0));
    }
    const size_t shade_lds = (size_t)h->always_doubles * sizeof(double);
    // the sub-pipelines' batches, enqueued round by round alternately so that neither stream runs ahead on the host
    uint64_t base[RT_MAX_SUB];
    for (int p = 0; p < nsub; p++) base[p] = sub_begin[p];
    for (;;) {
        bool any = false;
        uint64_t n[RT_MAX_SUB] = {};
        for (int p = 0; p < nsub; p++) {
            if (base[p] >= sub_begin[p + 1]) continue;
            any = true;
            const uint64_t left = sub_begin[p + 1] - base[p];
            n[p] = left < slots_cap[p] ? left : slots_cap[p];
            WfParams& w = wp[p];
            w.slots = (unsigned)n[p];
            w.item_base = (unsigned)base[p];
            hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(256), 0, ss[p], w.counts, (unsigned)iters + 2u, (unsigned)n[p]);
            w.iter = 0;
            unsigned igrid = (unsigned)((n[p] + 255) / 256);
            if (igrid > (unsigned)h->cu_count * 16u) igrid = (unsigned)h->cu_count * 16u;
            hipLaunchKernelGGL(init_kernel, dim3(igrid), dim3(256), 0, ss[p], w);
        }
        if (!any) break;
        for (int it = 0; it < iters; it++) {
            for (int p = 0; p < nsub; p++) {
                if (!n[p]) continue;
                WfParams& w = wp[p];
                w.iter = it;
                if (h->timing_detail) HIP_TRY(h, hipEventRecord(h->broad_ev[h->broad_ev_used++], ss[p]));
                if (use_grid) { hipLaunchKernelGGL(walk_kernel, dim3((unsigned)h->cu_count * 16u), dim3(256), 0, ss[p], w, h->grid); e = hipGetLastError(); }
                else if (h->broad_block == 1024) e = launch_broad<1024>(h, w, bgrid, ss[p]);
                else if (h->broad_block == 256) e = launch_broad<256>(h, w, bgrid, ss[p]);
                else e = launch_broad<512>(h, w, bgrid, ss[p]);
                if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("broad kernel launch: ") + hipGetErrorString(e));
                if (h->timing_detail) HIP_TRY(h, hipEventRecord(h->broad_ev[h->broad_ev_used++], ss[p]));
                hipLaunchKernelGGL(shade_kernel, dim3(sgrid * 256 / SHADE_BLOCK), dim3(SHADE_BLOCK), shade_lds, ss[p], w);
            }
        }
        e = hipGetLastError();
        if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("pipeline launch: ") + hipGetErrorString(e));
        for (int p = 0; p < nsub; p++) base[p] += n[p];
    }
    if (nsub > 1) {
        for (int p = 0; p < nsub; p++) {
            HIP_TRY(h, hipEventRecord(h->join_ev[p], ss[p]));
            HIP_TRY(h, hipStreamWaitEvent(st, h->join_ev[p], 0));
        }
    }
    HIP_TRY(h, hipEventRecord(h->ev1, st));
    h->events_pending = true;
    return RT_OK;
}

static int render_common(rt_handle h, int stream_mode, int tile_rows, int rank, int world, double* d_rgb,
                         int32_t* d_hit_ids, void* stream) {
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (stream_mode || h->mode == RT_MODE_SINGLE_KERNEL)
        return render_single_kernel(h, stream_mode, tile_rows, rank, world, d_rgb, d_hit_ids, st);
    return render_pipeline(h, tile_rows, rank, world, d_rgb, d_hit_ids, st);
}

int rt_render_rows_device(rt_handle h, int tile_rows, int rank, int world, double* d_rgb, int32_t* d_hit_ids, void* stream) {
    if (!h) return RT_ERR_ARG;
    if (!h->have_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render_rows_device: no scene uploaded");
    if (!d_rgb || tile_rows < 1 || world < 1 || rank < 0 || rank >= world) return fail(h, RT_ERR_ARG, "rt_render_rows_device: bad argument");
    return render_common(h, 0, tile_rows, rank, world, d_rgb, d_hit_ids, stream);
}

int rt_render_streams_device(rt_handle h, double* d_rgb, int32_t* d_hit_ids, void* stream) {
    if (!h) return RT_ERR_ARG;
    if (!h->have_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render_streams_device: no scene uploaded");
    if (!d_rgb) return fail(h, RT_ERR_ARG, "rt_render_streams_device: bad argument");
    return render_common(h, 1, 1, 0, 1, d_rgb, d_hit_ids, stream);
}

int rt_finish(rt_handle h, uint64_t* rays) {
    if (!h) return RT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->events_pending) return fail(h, RT_ERR_ARG, "rt_finish: no render in flight");
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    h->events_pending = false;
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    unsigned ctrl[8];
    HIP_TRY(h, hipMemcpy(ctrl, h->d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
    uint64_t r = (uint64_t)ctrl[2] | ((uint64_t)ctrl[3] << 32);
    h->timing.kernel_ms = ms;
    h->timing.broad_ms = 0.0;
    for (int i = 0; i + 1 < h->broad_ev_used; i += 2) {
        float bms = 0;
        if (hipEventElapsedTime(&bms, h->broad_ev[i], h->broad_ev[i + 1]) == hipSuccess) h->timing.broad_ms += bms;
    }
    h->timing.rays = r;
    h->timing.candidates = (uint64_t)ctrl[4] | ((uint64_t)ctrl[5] << 32);
    h->timing.overflow_rays = ctrl[6];
#if WALK_DEBUG
    {
        unsigned dbg[8];
        hipMemcpy(dbg, h->d_ctrl + 8, sizeof(dbg), hipMemcpyDeviceToHost);
        fprintf(stderr, "walk: non-finite %u  far %u  list full %u  cells visited %u  items tested %u  (rays %llu)\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], (unsigned long long)r);
    }
#endif
#if SHADE_PROFILE
    {
        unsigned long long prof[6];
        hipMemcpy(prof, h->d_ctrl + 16, sizeof(prof), hipMemcpyDeviceToHost);
        fprintf(stderr, "shade wave-cycles: loop-top %llu  load_state %llu  find_hit %llu  segment %llu  barriers %llu  emit %llu\n",
                prof[0], prof[5], prof[1], prof[2], prof[3], prof[4]);
    }
#endif
    h->timing.sphere_tests = r * h->n_spheres;
    h->timing.plane_tests = r * h->n_planes;
    h->timing.face_tests = r * h->n_faces;
    h->timing.face_vert_tests = r * h->n_face_verts;
    if (rays) *rays = r;
    if (ctrl[1] & ERR_REJECT_GUARD)
        return fail(h, RT_ERR_GUARD, "render: the glossy rejection loop did not accept within its guard at local pixel " + std::to_string(ctrl[7] - 1u) +
                                     " (a glossy surface hit from behind: the reference never terminates on this scene)");
    if (ctrl[1] & ERR_STACK) return fail(h, RT_ERR_GUARD, "render: bounce stack overflow (path longer than 2*max_ray_depth+2 vertices: no emitter in the scene?)");
    return RT_OK;
}

int rt_stream_rays(rt_handle h, uint64_t* out, int n) {
    if (!h || !out || n < 0) return RT_ERR_ARG;
    if (h->events_pending) return fail(h, RT_ERR_ARG, "rt_stream_rays: call rt_finish first");
    if (n > h->last_streams) return fail(h, RT_ERR_ARG, "rt_stream_rays: the last render had fewer streams");
    HIP_TRY(h, hipSetDevice(h->device));
    if (n) HIP_TRY(h, hipMemcpy(out, h->d_stream_rays, (size_t)n * 8, hipMemcpyDeviceToHost));
    return RT_OK;
}

static int render_host(rt_handle h, int stream_mode, double* rgb, uint64_t* rays, int32_t* hit_ids) {
    if (!h) return RT_ERR_ARG;
    if (!h->have_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render: no scene uploaded");
    if (!rgb) return fail(h, RT_ERR_ARG, "rt_render: rgb is null");
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    hipEventRecord(t0, 0);
    const size_t npx = (size_t)h->width * h->height;
    const size_t rb = npx * 3 * sizeof(double), hb = npx * h->spp * sizeof(int32_t);
    if (h->d_rgb_bytes < rb) { hipFree(h->d_rgb); h->d_rgb = nullptr; h->d_rgb_bytes = 0; HIP_TRY(h, hipMalloc((void**)&h->d_rgb, rb)); h->d_rgb_bytes = rb; }
    if (hit_ids && h->d_hits_bytes < hb) { hipFree(h->d_hits); h->d_hits = nullptr; h->d_hits_bytes = 0; HIP_TRY(h, hipMalloc((void**)&h->d_hits, hb)); h->d_hits_bytes = hb; }
    int rc = render_common(h, stream_mode, h->height, 0, 1, h->d_rgb, hit_ids ? h->d_hits : nullptr, nullptr);
    if (rc != RT_OK) return rc;
    rc = rt_finish(h, rays);
    HIP_TRY(h, hipMemcpy(rgb, h->d_rgb, rb, hipMemcpyDeviceToHost));
    if (hit_ids) HIP_TRY(h, hipMemcpy(hit_ids, h->d_hits, hb, hipMemcpyDeviceToHost));
    hipEventRecord(t1, 0);
    hipEventSynchronize(t1);
    float ms = 0;
    hipEventElapsedTime(&ms, t0, t1);
    h->timing.total_ms = ms;
    hipEventDestroy(t0); hipEventDestroy(t1);
    return rc;
}

int rt_render(rt_handle h, double* rgb, uint64_t* rays, int32_t* hit_ids) { return render_host(h, 0, rgb, rays, hit_ids); }

int rt_render_streams(rt_handle h, double* rgb, uint64_t* rays, int32_t* hit_ids) { return render_host(h, 1, rgb, rays, hit_ids); }

// the Gaussian taps and the two frame-sized buffers the blur passes alternate between (n pixels)
static int ensure_bloom(rt_handle h, size_t n) {
    if (!h->d_bloom_kern) {
        // the Gaussian of main.cpp:188-194, evaluated on the host with the C library's exp like the reference
        const double stddev = 3, denom = 2 * stddev * stddev;
        std::vector<double> k((size_t)BLOOM_K * BLOOM_K);
        for (int y = -BLOOM_R; y <= BLOOM_R; y++)
            for (int x = -BLOOM_R; x <= BLOOM_R; x++)
                k[(size_t)(y + BLOOM_R) * BLOOM_K + x + BLOOM_R] = exp(-(x * x + y * y) / denom) / denom / M_PI;
        HIP_TRY(h, hipMalloc((void**)&h->d_bloom_kern, k.size() * sizeof(double)));
        HIP_TRY(h, hipMemcpy(h->d_bloom_kern, k.data(), k.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h->bloom_doubles < n * 3) {
        for (int q = 0; q < 2; q++) { hipFree(h->d_bloom[q]); h->d_bloom[q] = nullptr; }
        h->bloom_doubles = 0;
        for (int q = 0; q < 2; q++) HIP_TRY(h, hipMalloc((void**)&h->d_bloom[q], n * 3 * sizeof(double)));
        h->bloom_doubles = n * 3;
    }
    return RT_OK;
}

int rt_expose_device(rt_handle h, const double* d_rgb, int width, int height, double exposure, int num_blurs,
                     int32_t* d_img, void* stream) {
    if (!h) return RT_ERR_ARG;
    if (!d_rgb || !d_img || width < 1 || height < 1 || num_blurs < 0) return fail(h, RT_ERR_ARG, "rt_expose_device: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)width * height;
    unsigned grid = (unsigned)((n + 255) / 256);
    if (grid > (unsigned)h->cu_count * 32u) grid = (unsigned)h->cu_count * 32u;
    if (num_blurs == 0) {
        hipLaunchKernelGGL(expose_pack_kernel, dim3(grid), dim3(256), 0, st, d_rgb, (const double*)nullptr, d_img, n, exposure);
        HIP_TRY(h, hipGetLastError());
        return RT_OK;
    }
    { const int rc = ensure_bloom(h, n); if (rc != RT_OK) return rc; }
    unsigned g3 = (unsigned)((n * 3 + 255) / 256);
    if (g3 > (unsigned)h->cu_count * 32u) g3 = (unsigned)h->cu_count * 32u;
    hipLaunchKernelGGL(expose_mask_kernel, dim3(g3), dim3(256), 0, st, d_rgb, h->d_bloom[0], n * 3, exposure);
    int cur = 0;
    const dim3 bgrid((width + BLOOM_T - 1) / BLOOM_T, (height + BLOOM_T - 1) / BLOOM_T, 3);
    for (int b = 0; b < num_blurs; b++) {
        hipLaunchKernelGGL(bloom_kernel, bgrid, dim3(BLOOM_T * BLOOM_T), 0, st, (const double*)h->d_bloom[cur], h->d_bloom[cur ^ 1],
                           width, height, (const double*)h->d_bloom_kern, (const int*)nullptr);
        cur ^= 1;
    }
    hipLaunchKernelGGL(expose_pack_kernel, dim3(grid), dim3(256), 0, st, d_rgb, (const double*)h->d_bloom[cur], d_img, n, exposure);
    HIP_TRY(h, hipGetLastError());
    return RT_OK;
}

int rt_expose_bits_device(rt_handle h, const double* d_rgb, int width, int rows, double exposure, uint8_t* d_bits, void* stream) {
    if (!h) return RT_ERR_ARG;
    if (!d_rgb || !d_bits || width < 1 || rows < 0) return fail(h, RT_ERR_ARG, "rt_expose_bits_device: bad argument");
    if (rows == 0) return RT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)width * rows;
    unsigned grid = (unsigned)((n + 255) / 256);
    if (grid > (unsigned)h->cu_count * 32u) grid = (unsigned)h->cu_count * 32u;
    hipLaunchKernelGGL(expose_bits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_rgb, d_bits, n, exposure);
    HIP_TRY(h, hipGetLastError());
    return RT_OK;
}

int rt_expose_rows_device(rt_handle h, const double* d_rgb, const uint8_t* d_bits_all, int width, int height, int tile_rows,
                          int rank, int world, int pad_rows, double exposure, int num_blurs, int32_t* d_img, void* stream) {
    if (!h) return RT_ERR_ARG;
    if (!d_rgb || !d_img || width < 1 || height < 1 || num_blurs < 0 || tile_rows < 1 || world < 1 || rank < 0 || rank >= world)
        return fail(h, RT_ERR_ARG, "rt_expose_rows_device: bad argument");
    const int rows = rt_rows_for_rank(height, tile_rows, rank, world);
    if (rows == 0) return RT_OK;
    // without a blur pass the bloom of a pixel is its own B0: nothing of another rank is needed
    if (num_blurs == 0) return rt_expose_device(h, d_rgb, width, rows, exposure, 0, d_img, stream);
    if (!d_bits_all) return fail(h, RT_ERR_ARG, "rt_expose_rows_device: the gathered bits are needed when num_blurs > 0");
    for (int r = 0; r < world; r++)
        if (rt_rows_for_rank(height, tile_rows, r, world) > pad_rows) return fail(h, RT_ERR_ARG, "rt_expose_rows_device: pad_rows is smaller than a rank's row count");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)width * height;
    { const int rc = ensure_bloom(h, n); if (rc != RT_OK) return rc; }
    // block-row lists, last pass first: the blocks holding the rank's own rows, then two more block rows either side per
    // earlier pass (a 16-row block's taps reach 20 rows); list 0 is where B0 is needed
    const int nblk = (height + BLOOM_T - 1) / BLOOM_T;
    const int key[5] = {height, tile_rows, rank, world, num_blurs};
    if (memcmp(key, h->bloom_list_key, sizeof(key)) != 0) {
        std::vector<char> need((size_t)nblk, 0);
        for (int lr = 0; lr < rows; lr++) need[(size_t)(((lr / tile_rows) * world + rank) * tile_rows + lr % tile_rows) / BLOOM_T] = 1;
        std::vector<std::vector<int>> lists((size_t)num_blurs + 1);
        for (int p = num_blurs; p >= 0; p--) {
            for (int b = 0; b < nblk; b++) if (need[b]) lists[p].push_back(b);
            std::vector<char> grown((size_t)nblk, 0);
            for (int b = 0; b < nblk; b++)
                if (need[b]) for (int k = b - 2; k <= b + 2; k++) if (k >= 0 && k < nblk) grown[k] = 1;
            need.swap(grown);
        }
        std::vector<int> flat;
        h->bloom_list_ofs.assign(1, 0);
        for (int p = 0; p <= num_blurs; p++) { flat.insert(flat.end(), lists[p].begin(), lists[p].end()); h->bloom_list_ofs.push_back((int)flat.size()); }
        HIP_TRY(h, hipStreamSynchronize(st));            // an earlier call on this stream may still read the old lists
        hipFree(h->d_bloom_lists); h->d_bloom_lists = nullptr;
        h->bloom_list_key[0] = -1;
        HIP_TRY(h, hipMalloc((void**)&h->d_bloom_lists, flat.size() * sizeof(int)));
        HIP_TRY(h, hipMemcpy(h->d_bloom_lists, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice));
        memcpy(h->bloom_list_key, key, sizeof(key));
    }
    const std::vector<int>& ofs = h->bloom_list_ofs;
    hipLaunchKernelGGL(expose_b0_blocks_kernel, dim3((unsigned)((width * BLOOM_T + 255) / 256), (unsigned)(ofs[1] - ofs[0])), dim3(256), 0, st,
                       d_bits_all, h->d_bloom[0], width, height, tile_rows, world, pad_rows, exposure, (const int*)h->d_bloom_lists);
    int cur = 0;
    for (int p = 1; p <= num_blurs; p++) {
        const dim3 bgrid((width + BLOOM_T - 1) / BLOOM_T, (unsigned)(ofs[p + 1] - ofs[p]), 3);
        hipLaunchKernelGGL(bloom_kernel, bgrid, dim3(BLOOM_T * BLOOM_T), 0, st, (const double*)h->d_bloom[cur], h->d_bloom[cur ^ 1],
                           width, height, (const double*)h->d_bloom_kern, (const int*)(h->d_bloom_lists + ofs[p]));
        cur ^= 1;
    }
    const size_t nl = (size_t)rows * width;
    unsigned grid = (unsigned)((nl + 255) / 256);
    if (grid > (unsigned)h->cu_count * 32u) grid = (unsigned)h->cu_count * 32u;
    hipLaunchKernelGGL(expose_pack_rows_kernel, dim3(grid), dim3(256), 0, st, d_rgb, (const double*)h->d_bloom[cur], d_img, width, rows,
                       tile_rows, rank, world, exposure);
    HIP_TRY(h, hipGetLastError());
    return RT_OK;
}

int rt_render_image(rt_handle h, int reference_order, int32_t* img, uint64_t* rays) {
    if (!h) return RT_ERR_ARG;
    if (!h->have_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render_image: no scene uploaded");
    if (!img) return fail(h, RT_ERR_ARG, "rt_render_image: img is null");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t npx = (size_t)h->width * h->height;
    const size_t rb = npx * 3 * sizeof(double), ib = npx * sizeof(int32_t);
    if (h->d_rgb_bytes < rb) { hipFree(h->d_rgb); h->d_rgb = nullptr; h->d_rgb_bytes = 0; HIP_TRY(h, hipMalloc((void**)&h->d_rgb, rb)); h->d_rgb_bytes = rb; }
    if (h->d_hits_bytes < ib) { hipFree(h->d_hits); h->d_hits = nullptr; h->d_hits_bytes = 0; HIP_TRY(h, hipMalloc((void**)&h->d_hits, ib)); h->d_hits_bytes = ib; }
    int rc = render_common(h, reference_order ? 1 : 0, h->height, 0, 1, h->d_rgb, nullptr, nullptr);
    if (rc != RT_OK) return rc;
    rc = rt_finish(h, rays);
    if (rc != RT_OK) return rc;
    rc = rt_expose_device(h, h->d_rgb, h->width, h->height, h->exposure, h->num_blurs, h->d_hits, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipMemcpy(img, h->d_hits, ib, hipMemcpyDeviceToHost));   // the packed image borrows the hit-id scratch buffer
    return RT_OK;
}

int rt_last_timing(rt_handle h, rt_timing* out) {
    if (!h || !out) return RT_ERR_ARG;
    *out = h->timing;
    return RT_OK;
}

int rt_selftest_math(rt_handle h, int fn, int n, const double* x, const double* y, double* out) {
    if (!h || n < 0 || !x || !out || fn < 0 || fn > 6) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    size_t b = (size_t)n * sizeof(double);
    HIP_TRY(h, hipMalloc((void**)&dx, b));
    HIP_TRY(h, hipMalloc((void**)&dy, b));
    HIP_TRY(h, hipMalloc((void**)&dout, b));
    HIP_TRY(h, hipMemcpy(dx, x, b, hipMemcpyHostToDevice));
    if (y) HIP_TRY(h, hipMemcpy(dy, y, b, hipMemcpyHostToDevice));
    else HIP_TRY(h, hipMemset(dy, 0, b));
    hipLaunchKernelGGL(selftest_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, fn, n, dx, dy, dout);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(out, dout, b, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dy); hipFree(dout);
    return RT_OK;
}

int rt_selftest_broad(rt_handle h, int n, const double* origins, const double* dirs, const double* centres,
                      const double* radii, int32_t* flags) {
    if (!h || n < 0 || !origins || !dirs || !centres || !radii || !flags) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // same record construction as rt_upload_scene, with T = 0 and cmax over the batch
    std::vector<float> rec((size_t)n * 4);
    double cmax = 0;
    for (int i = 0; i < n; i++) {
        const double ext = make_record(centres + 3 * (size_t)i, fabs(radii[i]), &rec[4 * (size_t)i]);
        if (ext > cmax) cmax = ext;
    }
    double *dO = nullptr, *dD = nullptr; float4* dR = nullptr; int* dF = nullptr;
    HIP_TRY(h, hipMalloc((void**)&dO, (size_t)n * 24));
    HIP_TRY(h, hipMalloc((void**)&dD, (size_t)n * 24));
    HIP_TRY(h, hipMalloc((void**)&dR, (size_t)n * 16));
    HIP_TRY(h, hipMalloc((void**)&dF, (size_t)n * 4));
    HIP_TRY(h, hipMemcpy(dO, origins, (size_t)n * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dD, dirs, (size_t)n * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dR, rec.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_broad_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, dO, dD, dR,
                       (float)(cmax * (1.0 + 1e-6)), 0.0, 0.0, 0.0, dF);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(flags, dF, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(dO); hipFree(dD); hipFree(dR); hipFree(dF);
    return RT_OK;
}

}  // extern "C"
