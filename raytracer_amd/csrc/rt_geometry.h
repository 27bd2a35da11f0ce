// Exact (double) arithmetic of the reference: vectors, RNG, closest-hit acceptance, the three intersect functions, sampling.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

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
// (double)g() / g.max(), main.cpp:255.  The divisor is a constant and the dividend an integer below 2^31, so the quotient
// is taken with the correctly rounded reciprocal y and one correction step, q = g y, q + (g - b q) y: three operations in
// place of a division's thirteen.  It IS the IEEE quotient for every one of the 2^31 - 2 possible draws
// (tools/check_rng_div.c compares them all).
DEV double rng_uniform(unsigned& s) {
    const double b = 2147483646.0, y = 0x1.00000004p-31;          // y = fl(1 / b)
    const double g = (double)rng_next(s);
    const double q = g * y;
    return __builtin_fma(__builtin_fma(-b, q, g), y, q);
}

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
// Face::intersect, geometry.cpp:160-168.  The loop ends at the first vertex that rejects the point: what comes back is the
// same -- a rejected point is a rejected point -- and every vertex is a read that the next one waits behind (a quad's four
// were four microseconds of a wave's time; requesting a vertex one ahead costs 15 spilled registers and gains nothing).
DEV d3 face_point(d3 a, d3 n, int num, const FaceV* v, d3 o, d3 d) {
    d3 p = plane_point(a, n, o, d);
    if (dnan(p.x)) return p;                 // plane missed: every comparison below would be false anyway
    bool miss = false;
    for (int i = 0; i < num && !miss; i++) {
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

// what one sample of the reflection lobe needs beyond the RNG state: the rotation that carries +z to the lobe's axis (unit
// axis n, angle theta: main.cpp:53-59,80-86), 1 / specular_exp, and whether the surface normal is (0, 0, 1) within 1e-4
// (the reference tests the NORMAL there, main.cpp:68,95)
struct LobeCtx { d3 n; double cost, sint, inv_exp; bool flat; };

// One sample (main.cpp:62-73, 89-100): two draws, the +z hemisphere sample, the rotation.  glossy: power-cosine lobe,
// else uniform hemisphere; both build their sample the same way and share this instruction stream, each lane keeping
// its own branch's operation sequence.
DEV d3 lobe_sample(bool glossy, const LobeCtx& c, unsigned& rng) {
    const double PI = 3.14159265358979323846;
    const double u1 = rng_uniform(rng);
    const double u = glossy ? rtm::powd(rtm::cosd(u1 * PI / 2.0), c.inv_exp) : u1;
    const double phi = rng_uniform(rng) * 2.0 * PI;
    const double sr = __builtin_sqrt(1.0 - u * u);
    double sp, cp;
    rtm::sincosd(phi, &sp, &cp);
    const d3 v = dmk(sr * cp, sr * sp, u);
    return c.flat ? v : rotate3(v, c.n, c.cost, c.sint);
}

// get_reflection_ray, main.cpp:47-104.  Returns the new direction (not re-normalised).  The glossy branch (:48-77,
// power-cosine lobe about the mirror direction, rejection loop until the sample points away from the surface) and the
// diffuse branch (:78-103, uniform hemisphere about the normal, no rejection).
// POOL = false: the rejection loop runs here, as in the reference.  POOL = true: ONE sample is drawn; if it is rejected,
// *pending is set, *ctx holds what further samples need, and the caller draws them (lobe_retry_pool): a wave then spends
// its lanes on the few rays that still draw instead of repeating the loop body for them alone.
template <bool POOL>
DEV d3 reflection_dir(d3 d, d3 normal, unsigned gloss_thr, double inv_exp, unsigned& rng, unsigned* err, unsigned item,
                      LobeCtx* ctx, bool* pending) {
    LobeCtx c;
    c.flat = __builtin_fabs(normal.z - 1.0) < 0.0001;
    c.inv_exp = inv_exp;
    const bool glossy = rng_next(rng) < gloss_thr;
    const d3 axis = glossy ? dneg(reflect_dir(d, normal)) : normal;
    const double theta = rtm::acosd(axis.z);
    c.n = dnorm(dcross(dmk(0.0, 0.0, 1.0), axis));
    rtm::sincosd(theta, &c.sint, &c.cost);
    d3 nd = dmk(0.0, 0.0, 0.0);
    for (int iter = 0;; iter++) {
        nd = lobe_sample(glossy, c, rng);
        if (!glossy || ddot(nd, normal) >= 0.0) break;
        if (POOL) { *ctx = c; *pending = true; break; }
        if (iter > REJECT_GUARD_ITERS) { atomicOr(err, ERR_REJECT_GUARD); atomicMax(err + 6, item + 1u); break; }
    }
    return nd;
}

// 16807^(2k) mod (2^31 - 1), k = 0 .. 64: the generator's state after 2k further draws is state * that (mod 2^31 - 1)
struct RngJump {
    unsigned v[65];
    constexpr RngJump() : v{} {
        unsigned long long x = 1;
        for (int k = 0; k <= 64; k++) { v[k] = (unsigned)x; x = x * 16807ull % 2147483647ull; x = x * 16807ull % 2147483647ull; }
    }
};
__device__ const RngJump RNG_JUMP = RngJump();
// s * m mod (2^31 - 1) for a state s in [1, 2^31 - 2] and a multiplier below 2^31, by folding the product twice
// (2^31 == 1 mod 2^31 - 1); the result is a state again (never 0: neither factor is a multiple of the prime)
DEV unsigned rng_mulmod(unsigned s, unsigned m) {
    const unsigned long long p = (unsigned long long)s * m;
    unsigned long long x = (p & 0x7fffffffull) + (p >> 31);
    x = (x & 0x7fffffffull) + (x >> 31);
    if (x >= 0x7fffffffull) x -= 0x7fffffffull;
    return (unsigned)x;
}
// lane index of the k-th set bit (k < popcount) of a wave-uniform 64-bit mask
DEV int nth_set_bit(unsigned long long mask, int k) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long low = mask & ((1ull << w) - 1ull);
        const int c = __popcll(low);
        const bool up = k >= c;
        k = up ? k - c : k;
        mask = up ? mask >> w : low;
        pos = up ? pos + w : pos;
    }
    return pos;
}

// The further samples of the rejection loop (main.cpp:61-76) for the rays of a wave whose first sample was rejected, drawn
// by the WHOLE wave (call it with every lane active).  Sample j of a ray is a pure function of the lobe and of the RNG
// state after 2j draws, and that state is the ray's state times a constant (RNG_JUMP): so the R rays that still draw are
// dealt K = 64 / R lanes each, lane a of a ray's share draws its sample number a, and the ray takes the first accepted one
// with the RNG state that goes with it -- the values the loop would have reached one sample at a time.  A ray none of whose
// K samples was accepted moves its state 2K draws on and goes round again.
//   need: this lane's ray still draws; c, normal: its lobe and surface normal; rng: its state after the rejected sample;
//   dir: receives the accepted sample.
DEV void lobe_retry_pool(bool need, const LobeCtx& c, d3 normal, unsigned& rng, d3& dir, unsigned* err, unsigned item) {
    const int lane = threadIdx.x & 63;
    unsigned long long mask = __ballot(need);
    int tries = 0;
    while (mask) {
        const int R = __popcll(mask), K = 64 / R;
        const int i = lane / K, a = lane - i * K;
        const bool work = i < R;
        if (work) { PROF_LANES(err - 1, 14); }
        const int src = nth_set_bit(mask, work ? i : 0);
        LobeCtx q;
        q.n = dmk(__shfl(c.n.x, src, 64), __shfl(c.n.y, src, 64), __shfl(c.n.z, src, 64));
        q.cost = __shfl(c.cost, src, 64); q.sint = __shfl(c.sint, src, 64); q.inv_exp = __shfl(c.inv_exp, src, 64);
        q.flat = __shfl((int)c.flat, src, 64) != 0;
        const d3 qn = dmk(__shfl(normal.x, src, 64), __shfl(normal.y, src, 64), __shfl(normal.z, src, 64));
        unsigned qr = rng_mulmod((unsigned)__shfl((int)rng, src, 64), RNG_JUMP.v[a]);
        const d3 nd = lobe_sample(true, q, qr);
        const bool ok = work && ddot(nd, qn) >= 0.0;
        const unsigned long long okm = __ballot(ok);
        // the owner: its share is lanes [rank K, rank K + K)
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        const unsigned long long share = need ? (okm >> (rank * K)) & (K == 64 ? ~0ull : ((1ull << K) - 1ull)) : 0ull;
        const bool got = share != 0ull;
        const int from = got ? rank * K + __builtin_ctzll(share) : lane;
        const d3 take = dmk(__shfl(nd.x, from, 64), __shfl(nd.y, from, 64), __shfl(nd.z, from, 64));
        const unsigned take_rng = (unsigned)__shfl((int)qr, from, 64);
        if (got) { dir = take; rng = take_rng; need = false; }
        else if (need) rng = rng_mulmod(rng, RNG_JUMP.v[K]);
        tries += K;
        if (tries > REJECT_GUARD_ITERS) {
            if (need) { atomicOr(err, ERR_REJECT_GUARD); atomicMax(err + 6, item + 1u); }
            break;
        }
        mask = __ballot(need);
    }
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

// (int)x as the reference's compiler converts it (cvttsd2si): truncation, 0x80000000 when out of range or NaN
DEV int to_int_x86(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000u;
    return (int)v;
}

// Plane::get_material, geometry.cpp:106-116.  A hit further than 2^31 * 5 units from the plane's point overflows the
// reference's int conversion, which on its x86 target yields INT_MIN (an even number): the checker is uniform out there.
DEV int plane_material(double ax, double az, int mat0, int mat1, d3 p) {
    double vx = p.x - ax, vz = p.z - az;
    int x = to_int_x86(vx / 5.0) % 2;
    int z = to_int_x86(vz / 5.0) % 2;
    if (vx < 0.0) x = -x;
    if (vz < 0.0) z = -z;
    int m = (x == z) ? mat0 : mat1;
    if ((vx < 0.0) != (vz < 0.0)) m = (x == z) ? mat1 : mat0;
    return m;
}

}  // namespace
