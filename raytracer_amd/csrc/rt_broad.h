// Broad phase: the conservative FP32 ray / bounding-sphere test and its per-ray constants.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

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
// "wide" bounds, for the evaluation on the matrix cores (rt_kernel_mfma.h): the dot products drop partial products of at
// most 3 * 2^-24 of the largest and are accumulated over up to 32 terms in an order the hardware does not state, each
// addition rounding by at most u times the running sum <= S (resp. S^2):  Eh = 64 u S,  E0 = 96 u S^2.  delta is chosen
// near Eh / |hb| for typical rays, which minimises delta hb^2 + Eh^2 / delta.
constexpr float BROAD_WIDE_EH = 64.0f, BROAD_WIDE_E0 = 96.0f;
constexpr float BROAD_WIDE_DELTA = 6.0e-6f;    // the applied scale 1.0000035^2 - 1 = 7.0e-6 is larger
constexpr float BROAD_WIDE_SCALE = 1.0000035f;

// Square root for the bounds of the conservative tests: the hardware's own (v_sqrt_f32, within one unit in the last place)
// instead of the correctly rounded one, which is a dozen instructions more.  Every use leaves room for that unit: a factor
// 1.0001 on |o|, bounds with a threefold margin over the term-by-term count.
DEV float bound_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }

struct RayRec {
    float4 a;   // -d' (xyz), od'
    float4 b;   // 2 o (xyz), thr
    float4 c;   // behind thresholds: hb' > c.x and nq < c.y;  wide records: z, w = the constants of mf_behind (else +inf, 0)
};

DEV RayRec broad_setup(d3 o, d3 d, const double* T, float cmax, bool wide = false) {
    RayRec r;
    const float scale = wide ? BROAD_WIDE_SCALE : BROAD_SCALE, delta = wide ? BROAD_WIDE_DELTA : BROAD_DELTA;
    double px = o.x - T[0], py = o.y - T[1], pz = o.z - T[2];
    bool far = false, moved = false;
    double tmove = 0.0;
    if (wide) {
        // The matrix-core test looks at the ray's whole line and has no use for the origin as such, so the origin is
        // moved along the line to the point nearest the scene's centre: |o| and with it the error bounds shrink, and the
        // constants of a ray that starts far outside the scene (a bounce off the floor plane a long way out) stay inside the
        // f16 range.  The move is done in double; what it adds to the line's position, a few 2^-53 |o|, is covered by the
        // slack of Eh while |o| stays below 2^20 scene radii -- beyond that the ray is handed over as not ordinary.
        // Every record lies within cmax of the centre: a line further out than 2 cmax misses them all and may be
        // replaced by any other such line (the one at 2 cmax: same keep bits, ordinary magnitudes).
        const double lim = 1048576.0 * ((double)cmax + 1.0);
        far = !(__builtin_fabs(px) < lim) | !(__builtin_fabs(py) < lim) | !(__builtin_fabs(pz) < lim);
        const double dd = d.x * d.x + d.y * d.y + d.z * d.z;
        if (!far && dd > 1e-300 && dd < 1e300) {
            const double t = -(px * d.x + py * d.y + pz * d.z) / dd;
            px += t * d.x; py += t * d.y; pz += t * d.z;
            tmove = t; moved = true;
            const double D2 = px * px + py * py + pz * pz, out = 2.0 * (double)cmax;
            if (D2 > out * out) { const double k = out / __builtin_sqrt(D2); px *= k; py *= k; pz *= k; moved = false; }   // another line: the origin is not on it
        }
    }
    const float ox = (float)px, oy = (float)py, oz = (float)pz;
    const float fx = (float)d.x * scale, fy = (float)d.y * scale, fz = (float)d.z * scale;
    const double od = (double)fx * ox + (double)fy * oy + (double)fz * oz;
    const double o2 = (double)ox * ox + (double)oy * oy + (double)oz * oz;
    const float S = bound_sqrtf((float)o2) * 1.0001f + cmax;
    const float Eh = (wide ? BROAD_WIDE_EH : 14.0f) * F32_U * S;
    const float E0 = (wide ? BROAD_WIDE_E0 : 24.0f) * F32_U * S * S;
    r.a = make_float4(-fx, -fy, -fz, (float)od);
    const double thr = o2 - (double)E0 - (double)Eh * Eh * (1.0 + 1.0 / (double)delta);
    // rounded down, and strictly below: the pipeline keeps a record iff thr - s is negative
    const float thr_f = (float)thr - __builtin_fabsf((float)thr) * (4.0f * F32_U) - 1e-30f;
    r.b = make_float4(2.0f * ox, 2.0f * oy, 2.0f * oz, thr_f);
    const float bc = (float)(o2 - (double)E0) - __builtin_fabsf((float)o2) * (4.0f * F32_U);
    // Where the origin went.  The matrix-core test sees hb' = d'.(o' - c) at the MOVED origin o'; at the ray's true origin
    // o = o' - tmove d it is hb' - tmove (d'.d).  A sphere whose line parameters (-B +- sqrt(disc), B = d.(o - c)) are both
    // negative lies behind the origin and cannot be hit: B > 0 and B^2 > disc.  The kernel knows disc from above
    // (thr - s >= disc: the bound that makes the keep test conservative) and B from below:
    //     hbt = hb_measured - tmv <= d'.(o - c) - Eh,    tmv = tmove (d'.d) rounded up + 2 Eh,
    // so hbt > 0 and hbt^2 > (s - thr) + mr imply it, mr covering the scale of d' (hb'^2 <= (1 + 7.1e-6) B^2 (1 + 2^-22)), the
    // rounding of the two fused operations and of s - thr (each below 2^-22 of 4 S^2).  Without a moved origin: never.
    float tmv = __builtin_inff(), mr = 0.0f;
    if (wide && moved) {
        const double tmd = tmove * ((double)fx * d.x + (double)fy * d.y + (double)fz * d.z);
        const float tf = (float)tmd;
        tmv = tf + __builtin_fabsf(tf) * (4.0f * F32_U) + 2.0f * Eh + 1e-30f;
        mr = 6.0e-5f * S * S;
    }
    r.c = make_float4(Eh * 1.00001f, bc, tmv, mr);
    // A ray whose float image is not of ordinary size -- NaN, infinite, or so large that the squares above leave the float
    // range (inf - inf is a NaN of either sign, and the keep rule reads a sign) -- gets constants under which every record
    // is kept and none is "behind": s = the record's own fourth word, thr = -inf.  The exact test then sees every solid.
    const float lim = 1e15f;
    const bool ordinary = (__builtin_fabsf(ox) < lim) & (__builtin_fabsf(oy) < lim) & (__builtin_fabsf(oz) < lim) &
                          (__builtin_fabsf(fx) < lim) & (__builtin_fabsf(fy) < lim) & (__builtin_fabsf(fz) < lim) & !far;
    if (!ordinary) {
        r.a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        r.b = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
        r.c = make_float4(__builtin_inff(), 0.0f, __builtin_inff(), 0.0f);
    }
    return r;
}

// s and hb' of one record (cx, cy, cz, R^2 - |c|^2) for one ray; nq is returned for the behind test
DEV float broad_s(const float4 rec, const float4 ra, const float4 rb, float& hb, float& nq) {
    hb = __builtin_fmaf(ra.z, rec.z, __builtin_fmaf(ra.y, rec.y, __builtin_fmaf(ra.x, rec.x, ra.w)));
    nq = __builtin_fmaf(rb.z, rec.z, __builtin_fmaf(rb.y, rec.y, __builtin_fmaf(rb.x, rec.x, rec.w)));
    return __builtin_fmaf(hb, hb, nq);
}
DEV bool broad_behind(float hb, float nq, const float4 rc) { return hb > rc.x && nq < rc.y; }

}  // namespace
