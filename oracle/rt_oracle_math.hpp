// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into the product library.
//
// Portable transcendental functions of the oracle's "device arithmetic" policy.
//
// The reference calls libm (sin, cos, acos, pow: /root/reference/src/main.cpp:53,
// 58-59,62-66,80,85-86; /root/reference/src/geometry.cpp:201).  A GPU cannot
// reproduce glibc's results bit for bit, so the device policy replaces these four
// functions -- and nothing else -- by the ones below, which use only operations
// that IEEE-754 defines exactly (double + - * / sqrt fma, int<->double
// conversion, integer bit operations).  Double in, double out, a few ulp from
// the true value (pow: ~1e-13 relative for exponents around 1000).  The
// specification of each function is its operation sequence; the device library
// carries its own copy of the same sequence and the parity tests compare the two
// bit for bit.
#pragma once
#include <cstdint>
#include <cstring>

namespace orm {

#include "rt_oracle_coeffs.inc"

static inline double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline uint64_t d2u(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; std::memcpy(&x, &u, 8); return x; }
static inline double abs_d(double x) { return u2d(d2u(x) & 0x7fffffffffffffffull); }
static inline double qnan_d() { return u2d(0x7ff8000000000000ull); }
static inline double inf_d() { return u2d(0x7ff0000000000000ull); }
static inline bool isnan_d(double x) { return (d2u(x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }

// sin and cos of |x| < 2^20: quadrant reduction with a three-part pi/2, Taylor polynomials.
static inline void sincos_d(double x, double* sn, double* cs) {
    if (!(abs_d(x) < 1048576.0)) { *sn = qnan_d(); *cs = qnan_d(); return; }
    double t = x * ORM_TWO_OVER_PI;
    int k = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = fma_d(-kd, ORM_PIO2_1, x);
    r = fma_d(-kd, ORM_PIO2_2, r);
    r = fma_d(-kd, ORM_PIO2_3, r);
    double r2 = r * r;
    double ps = ORM_SIN[8];
    for (int i = 7; i >= 0; i--) ps = fma_d(ps, r2, ORM_SIN[i]);
    double s = fma_d(r * r2, ps, r);
    double pc = ORM_COS[9];
    for (int i = 8; i >= 1; i--) pc = fma_d(pc, r2, ORM_COS[i]);
    double c = fma_d(r2 * r2, pc, fma_d(r2, -0.5, 1.0));
    switch (k & 3) {
        case 0: *sn = s;  *cs = c;  break;
        case 1: *sn = c;  *cs = -s; break;
        case 2: *sn = -s; *cs = -c; break;
        default: *sn = -c; *cs = s; break;
    }
}

static inline double p_sin(double x) { double s, c; sincos_d(x, &s, &c); return s; }
static inline double p_cos(double x) { double s, c; sincos_d(x, &s, &c); return c; }

static inline double asin_poly_d(double x) {  // |x| <= 0.5
    double x2 = x * x;
    double p = ORM_ASIN[29];
    for (int i = 28; i >= 0; i--) p = fma_d(p, x2, ORM_ASIN[i]);
    return fma_d(x * x2, p, x);
}

static inline double p_acos(double x) {
    double ax = abs_d(x);
    if (ax <= 0.5) return (ORM_PIO2 - asin_poly_d(x)) + ORM_PIO2_LOW;
    double z = (1.0 - ax) * 0.5;          // negative for |x| > 1: sqrt gives NaN
    double s = __builtin_sqrt(z);
    double a = 2.0 * asin_poly_d(s);
    return (x > 0.0) ? a : ((ORM_PI - a) + ORM_PI_LOW);
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
    double p = ORM_LOG[12];
    for (int i = 11; i >= 0; i--) p = fma_d(p, s2, ORM_LOG[i]);
    double two_s = 2.0 * s;
    double lnm = fma_d(two_s * s2, p, two_s);
    return fma_d(lnm, ORM_INV_LN2, (double)e);
}

// 2^p for -1100 <= p <= 1030 (results below the normal range are rounded once, by the last product).
static inline double exp2_d(double p) {
    int n = (int)(p + (p >= 0.0 ? 0.5 : -0.5));
    double f = p - (double)n;
    double z = f * ORM_LN2;
    double e = ORM_EXP[14];
    for (int i = 13; i >= 0; i--) e = fma_d(e, z, ORM_EXP[i]);
    double r = fma_d(z * z, e, z) + 1.0;
    if (n > 1000) return (r * u2d((uint64_t)(n - 500 + 1023) << 52)) * u2d((uint64_t)(500 + 1023) << 52);
    if (n < -1000) return (r * u2d((uint64_t)(n + 500 + 1023) << 52)) * u2d((uint64_t)(-500 + 1023) << 52);
    return r * u2d((uint64_t)(n + 1023) << 52);
}

// pow with the IEEE special cases the renderer can reach.  Zero results carry no sign
// information (pow(-0, odd) gives +0 here); subnormal bases are treated as zero.
static inline double p_pow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (x == 1.0) return 1.0;
    if (isnan_d(x) || isnan_d(y)) return qnan_d();
    double ax = abs_d(x), ay = abs_d(y);
    bool y_int, y_odd;
    if (ay >= 9007199254740992.0) { y_int = true; y_odd = false; }   // also +-inf: treated as even
    else if (ay >= 2147483648.0) { double h = y * 0.5; double fl = (double)(long long)h; y_int = ((double)(long long)y == y); y_odd = y_int && (fl * 2.0 != y); }
    else { int yi = (int)y; y_int = ((double)yi == y); y_odd = (yi & 1) != 0; }
    double sign = 1.0;
    if (x < 0.0) {
        if (!y_int) return qnan_d();
        if (y_odd) sign = -1.0;
    }
    if (ax == 1.0) return sign;                                      // x == -1
    double L;
    if (ax < 2.2250738585072014e-308) L = -inf_d();
    else if (ax == inf_d()) L = inf_d();
    else L = log2_d(ax);
    double P = y * L;
    double r;
    if (P > 1030.0) r = inf_d();
    else if (P < -1100.0) r = 0.0;
    else r = exp2_d(P);
    return sign * r;
}

}  // namespace orm
