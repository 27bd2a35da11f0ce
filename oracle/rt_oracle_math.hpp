// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into the product library.
//
// Portable transcendental functions of the oracle's "device arithmetic" policy.
//
// The reference calls libm (sin, cos, acos, pow: /root/reference/src/main.cpp:53,
// 58-59,62-66,80,85-86; /root/reference/src/geometry.cpp:201).  A GPU cannot
// reproduce glibc's results bit for bit, so the device policy replaces them by
// the functions below, which use only operations that IEEE-754 defines exactly
// (+ - * / fma on doubles, float<->double conversion, float sqrt, integer bit
// operations).  Float in, float out, double inside: the result is the float
// nearest to a double value whose error is below 1e-13 relative, i.e. the
// correctly rounded float except in rare near-tie cases.  The specification of
// each function is its operation sequence; the device library carries its own
// copy of the same sequence and the parity tests compare the two bit for bit.
#pragma once
#include <cstdint>
#include <cstring>

namespace orm {

#include "rt_oracle_coeffs.inc"

static inline double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline uint64_t d2u(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; std::memcpy(&x, &u, 8); return x; }
static inline uint32_t f2u(float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; std::memcpy(&x, &u, 4); return x; }
static inline double abs_d(double x) { return u2d(d2u(x) & 0x7fffffffffffffffull); }
static inline float qnan_f() { return u2f(0x7fc00000u); }
static inline bool isnan_f(float x) { return (f2u(x) & 0x7fffffffu) > 0x7f800000u; }

// sin and cos of a double |x| < 2^20 by quadrant reduction and Taylor polynomials.
static inline void sincos_d(double x, double* sn, double* cs) {
    double t = x * ORM_TWO_OVER_PI;
    long k = (long)(t + (t >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = fma_d(-kd, ORM_PIO2_HI, x);
    r = fma_d(-kd, ORM_PIO2_LO, r);
    double r2 = r * r;
    double ps = ORM_SIN[5];
    for (int i = 4; i >= 0; i--) ps = fma_d(ps, r2, ORM_SIN[i]);
    double s = fma_d(r * r2, ps, r);
    double pc = ORM_COS[6];
    for (int i = 5; i >= 0; i--) pc = fma_d(pc, r2, ORM_COS[i]);
    double c = fma_d(r2, pc, 1.0);
    switch (k & 3) {
        case 0: *sn = s;  *cs = c;  break;
        case 1: *sn = c;  *cs = -s; break;
        case 2: *sn = -s; *cs = -c; break;
        default: *sn = -c; *cs = s; break;
    }
}

static inline float p_sin(float x) {
    double xd = (double)x;
    if (!(abs_d(xd) < 1048576.0)) return qnan_f();
    double s, c;
    sincos_d(xd, &s, &c);
    return (float)s;
}

static inline float p_cos(float x) {
    double xd = (double)x;
    if (!(abs_d(xd) < 1048576.0)) return qnan_f();
    double s, c;
    sincos_d(xd, &s, &c);
    return (float)c;
}

static inline double asin_poly_d(double x) {  // |x| <= 0.5
    double x2 = x * x;
    double p = ORM_ASIN[18];
    for (int i = 17; i >= 0; i--) p = fma_d(p, x2, ORM_ASIN[i]);
    return fma_d(x * x2, p, x);
}

// sqrt of a double in [0, 0.25] from a float seed and two Newton steps; every
// operation is exactly rounded, so the result is reproducible (not necessarily
// the correctly rounded square root).  Negative input gives NaN.
static inline double sqrt_newton_d(double z) {
    if (z == 0.0) return 0.0;
    double y = (double)__builtin_sqrtf((float)z);
    y = 0.5 * (y + z / y);
    y = 0.5 * (y + z / y);
    return y;
}

static inline float p_acos(float x) {
    double xd = (double)x;
    double ax = abs_d(xd);
    double r;
    if (ax <= 0.5) {
        r = ORM_PIO2 - asin_poly_d(xd);
    } else {
        double z = (1.0 - ax) * 0.5;
        double s = sqrt_newton_d(z);
        double a = 2.0 * asin_poly_d(s);
        r = (xd > 0.0) ? a : (ORM_PI - a);
    }
    return (float)r;
}

// log2 of a positive, finite, normal double.
static inline double log2_d(double a) {
    uint64_t b = d2u(a);
    int e = (int)(b >> 52) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > ORM_SQRT2) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double s2 = s * s;
    double p = ORM_LOG[10];
    for (int i = 9; i >= 0; i--) p = fma_d(p, s2, ORM_LOG[i]);
    double two_s = 2.0 * s;
    double lnm = fma_d(two_s * s2, p, two_s);
    return fma_d(lnm, ORM_INV_LN2, (double)e);
}

// 2^p for -160 <= p <= 130.
static inline double exp2_d(double p) {
    long n = (long)(p + (p >= 0.0 ? 0.5 : -0.5));
    double f = p - (double)n;
    double z = f * ORM_LN2;
    double e = ORM_EXP[12];
    for (int i = 11; i >= 0; i--) e = fma_d(e, z, ORM_EXP[i]);
    double r = fma_d(z * z, e, z) + 1.0;
    return r * u2d((uint64_t)(n + 1023) << 52);
}

// pow with the IEEE special cases the renderer can reach.  Zero results carry no
// sign information (pow(-0, odd) gives +0 here).
static inline float p_pow(float xf, float yf) {
    if (yf == 0.0f) return 1.0f;
    if (xf == 1.0f) return 1.0f;
    if (isnan_f(xf) || isnan_f(yf)) return qnan_f();
    double x = (double)xf, y = (double)yf;
    double ax = abs_d(x);
    double ay = abs_d(y);
    bool y_int, y_odd;
    if (ay >= 16777216.0) { y_int = true; y_odd = false; }   // also +-inf: treated as even
    else { long yi = (long)y; y_int = ((double)yi == y); y_odd = (yi & 1) != 0; }
    double sign = 1.0;
    if (x < 0.0) {
        if (!y_int) return qnan_f();
        if (y_odd) sign = -1.0;
    }
    if (ax == 1.0) return (float)sign;                        // x == -1
    double L;
    if (ax == 0.0) L = -__builtin_inf();
    else if (ax == __builtin_inf()) L = __builtin_inf();
    else L = log2_d(ax);
    double P = y * L;
    double r;
    if (P > 130.0) r = __builtin_inf();
    else if (P < -160.0) r = 0.0;
    else r = exp2_d(P);
    return (float)(sign * r);
}

}  // namespace orm
