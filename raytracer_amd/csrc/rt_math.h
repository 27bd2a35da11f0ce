// Bit-reproducible transcendental functions for the render kernels.
//
// The reference renderer computes in double and calls libm's sin, cos, acos and
// pow (/root/reference/src/main.cpp:53,58-59,62-66,80,85-86 and
// /root/reference/src/geometry.cpp:201).  Everything else it does (+ - * / sqrt
// on doubles) is defined exactly by IEEE-754 and a GPU reproduces it bit for bit;
// those four functions are not, so the kernels use the ones below instead.  Each
// is a fixed sequence of exactly defined operations -- double add/mul/div/sqrt/
// fma, int<->double conversion, integer bit operations -- so the same input gives
// the same bits on gfx950 and on any IEEE host.  Accuracy: sin <= 2 ulp, cos and
// acos <= 1 ulp, pow <= 1 ulp for |y| <= 1 and about 1e-13 relative for exponents
// near 1000 (measured against glibc).
//
// The translation unit that includes this header must be compiled with
// -ffp-contract=off: a fused multiply-add may appear only where fma is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTM_FN __host__ __device__ __forceinline__
#ifndef RTM_BIG
#define RTM_BIG __host__ __device__ __attribute__((noinline))
#endif

namespace rtm {

#include "rt_math_coeffs.inc"

RTM_FN double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
RTM_FN double bits2d(uint64_t u) { union { uint64_t u; double d; } x; x.u = u; return x.d; }
RTM_FN uint64_t d2bits(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }
RTM_FN double absd(double x) { return bits2d(d2bits(x) & 0x7fffffffffffffffull); }
RTM_FN double nand() { return bits2d(0x7ff8000000000000ull); }
RTM_FN double infd() { return bits2d(0x7ff0000000000000ull); }
RTM_FN bool isnand(double x) { return (d2bits(x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }

// sin and cos together: reduce by multiples of pi/2 (three-constant Cody-Waite), Taylor
// polynomials on [-pi/4, pi/4], quadrant fix-up.  NaN for |x| >= 2^20.
struct SinCos { double s, c; };
RTM_BIG SinCos sincos2(double x) {
    SinCos out;
    if (!(absd(x) < 1048576.0)) { out.s = nand(); out.c = nand(); return out; }
    double t = x * RTM_TWO_OVER_PI;
    int k = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = fmad(-kd, RTM_PIO2_1, x);
    r = fmad(-kd, RTM_PIO2_2, r);
    r = fmad(-kd, RTM_PIO2_3, r);
    double r2 = r * r;
    double ps = RTM_SIN_8;
    ps = fmad(ps, r2, RTM_SIN_7); ps = fmad(ps, r2, RTM_SIN_6); ps = fmad(ps, r2, RTM_SIN_5);
    ps = fmad(ps, r2, RTM_SIN_4); ps = fmad(ps, r2, RTM_SIN_3); ps = fmad(ps, r2, RTM_SIN_2);
    ps = fmad(ps, r2, RTM_SIN_1); ps = fmad(ps, r2, RTM_SIN_0);
    double s = fmad(r * r2, ps, r);
    double pc = RTM_COS_9;
    pc = fmad(pc, r2, RTM_COS_8); pc = fmad(pc, r2, RTM_COS_7); pc = fmad(pc, r2, RTM_COS_6);
    pc = fmad(pc, r2, RTM_COS_5); pc = fmad(pc, r2, RTM_COS_4); pc = fmad(pc, r2, RTM_COS_3);
    pc = fmad(pc, r2, RTM_COS_2); pc = fmad(pc, r2, RTM_COS_1);
    double c = fmad(r2 * r2, pc, fmad(r2, -0.5, 1.0));
    int q = k & 3;
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    out.s = so;
    out.c = co;
    return out;
}

RTM_FN void sincosd(double x, double* sn, double* cs) { const SinCos r = sincos2(x); *sn = r.s; *cs = r.c; }
RTM_FN double sind(double x) { return sincos2(x).s; }
RTM_FN double cosd(double x) { return sincos2(x).c; }

RTM_FN double asin_poly(double x) {  // |x| <= 0.5: x + x^3 * (30-term Taylor tail in x^2)
    double x2 = x * x;
    double p = RTM_ASIN_29;
    p = fmad(p, x2, RTM_ASIN_28); p = fmad(p, x2, RTM_ASIN_27); p = fmad(p, x2, RTM_ASIN_26);
    p = fmad(p, x2, RTM_ASIN_25); p = fmad(p, x2, RTM_ASIN_24); p = fmad(p, x2, RTM_ASIN_23);
    p = fmad(p, x2, RTM_ASIN_22); p = fmad(p, x2, RTM_ASIN_21); p = fmad(p, x2, RTM_ASIN_20);
    p = fmad(p, x2, RTM_ASIN_19); p = fmad(p, x2, RTM_ASIN_18); p = fmad(p, x2, RTM_ASIN_17);
    p = fmad(p, x2, RTM_ASIN_16); p = fmad(p, x2, RTM_ASIN_15); p = fmad(p, x2, RTM_ASIN_14);
    p = fmad(p, x2, RTM_ASIN_13); p = fmad(p, x2, RTM_ASIN_12); p = fmad(p, x2, RTM_ASIN_11);
    p = fmad(p, x2, RTM_ASIN_10); p = fmad(p, x2, RTM_ASIN_9); p = fmad(p, x2, RTM_ASIN_8);
    p = fmad(p, x2, RTM_ASIN_7); p = fmad(p, x2, RTM_ASIN_6); p = fmad(p, x2, RTM_ASIN_5);
    p = fmad(p, x2, RTM_ASIN_4); p = fmad(p, x2, RTM_ASIN_3); p = fmad(p, x2, RTM_ASIN_2);
    p = fmad(p, x2, RTM_ASIN_1); p = fmad(p, x2, RTM_ASIN_0);
    return fmad(x * x2, p, x);
}

// acos: pi/2 - asin(x) for |x| <= 1/2, else 2 asin(sqrt((1-|x|)/2)) reflected for x < 0.
// |x| > 1 makes the sqrt argument negative, hence NaN.
RTM_BIG double acosd(double x) {
    // one polynomial evaluation serves both ranges (the lanes of a wave usually hold both)
    const double ax = absd(x);
    const bool small = ax <= 0.5;
    const double z = (1.0 - ax) * 0.5;
    const double t = small ? x : __builtin_sqrt(z);
    const double as = asin_poly(t);
    if (small) return (RTM_PIO2 - as) + RTM_PIO2_LOW;
    const double a = 2.0 * as;
    return (x > 0.0) ? a : ((RTM_PI - a) + RTM_PI_LOW);
}

// log2 of a positive finite normal double: split exponent, fold the mantissa into
// [sqrt(1/2), sqrt(2)], atanh series in s = f/(2+f)
RTM_FN double log2d(double a) {
    uint64_t b = d2bits(a);
    int e = (int)(b >> 52) - 1023;
    double m = bits2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > RTM_SQRT2) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double s2 = s * s;
    double p = RTM_LOG_12;
    p = fmad(p, s2, RTM_LOG_11); p = fmad(p, s2, RTM_LOG_10); p = fmad(p, s2, RTM_LOG_9);
    p = fmad(p, s2, RTM_LOG_8); p = fmad(p, s2, RTM_LOG_7); p = fmad(p, s2, RTM_LOG_6);
    p = fmad(p, s2, RTM_LOG_5); p = fmad(p, s2, RTM_LOG_4); p = fmad(p, s2, RTM_LOG_3);
    p = fmad(p, s2, RTM_LOG_2); p = fmad(p, s2, RTM_LOG_1); p = fmad(p, s2, RTM_LOG_0);
    double two_s = 2.0 * s;
    double lnm = fmad(two_s * s2, p, two_s);
    return fmad(lnm, RTM_INV_LN2, (double)e);
}

RTM_FN double pow2i(int n) { return bits2d((uint64_t)(int64_t)(n + 1023) << 52); }

// 2^p for -1100 <= p <= 1030: nearest-integer split, Taylor series of exp on the fraction;
// results outside the normal range are produced by one final rounding multiply.
RTM_FN double exp2d(double p) {
    int n = (int)(p + (p >= 0.0 ? 0.5 : -0.5));
    double f = p - (double)n;
    double z = f * RTM_LN2;
    double e = RTM_EXP_14;
    e = fmad(e, z, RTM_EXP_13); e = fmad(e, z, RTM_EXP_12); e = fmad(e, z, RTM_EXP_11);
    e = fmad(e, z, RTM_EXP_10); e = fmad(e, z, RTM_EXP_9); e = fmad(e, z, RTM_EXP_8);
    e = fmad(e, z, RTM_EXP_7); e = fmad(e, z, RTM_EXP_6); e = fmad(e, z, RTM_EXP_5);
    e = fmad(e, z, RTM_EXP_4); e = fmad(e, z, RTM_EXP_3); e = fmad(e, z, RTM_EXP_2);
    e = fmad(e, z, RTM_EXP_1); e = fmad(e, z, RTM_EXP_0);
    double r = fmad(z * z, e, z) + 1.0;
    if (n > 1000) return (r * pow2i(n - 500)) * pow2i(500);
    if (n < -1000) return (r * pow2i(n + 500)) * pow2i(-500);
    return r * pow2i(n);
}

// pow(x, y) = sign * 2^(y log2|x|) with pow's special cases: y == 0 or x == 1 give 1, NaN
// propagates, a negative base needs an integral y (odd y keeps the sign), x == -1 gives +-1,
// zero (and subnormal) and infinite bases go through log2 = -+inf.  A zero result is always +0.
RTM_BIG double powd(double x, double y) {
    if (y == 0.0) return 1.0;
    if (x == 1.0) return 1.0;
    if (isnand(x) || isnand(y)) return nand();
    double ax = absd(x), ay = absd(y);
    bool y_int, y_odd;
    if (ay >= 9007199254740992.0) { y_int = true; y_odd = false; }
    else if (ay >= 2147483648.0) {
        double h = y * 0.5;
        double fl = (double)(long long)h;
        y_int = ((double)(long long)y == y);
        y_odd = y_int && (fl * 2.0 != y);
    } else { int yi = (int)y; y_int = ((double)yi == y); y_odd = (yi & 1) != 0; }
    double sign = 1.0;
    if (x < 0.0) {
        if (!y_int) return nand();
        if (y_odd) sign = -1.0;
    }
    if (ax == 1.0) return sign;
    double L;
    if (ax < 2.2250738585072014e-308) L = -infd();
    else if (ax == infd()) L = infd();
    else L = log2d(ax);
    double P = y * L;
    double r;
    if (P > 1030.0) r = infd();
    else if (P < -1100.0) r = 0.0;
    else r = exp2d(P);
    return sign * r;
}

}  // namespace rtm
