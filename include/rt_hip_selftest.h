/* rt_hip_selftest.h -- self-test and inspection entry points of librt_hip.so.
 *
 * Not part of the render path: the parity suite uses them to look at single pieces of the device arithmetic (the math
 * functions, the conservative FP32 / matrix-core broad phase) and at host-side tables the library derives from a scene
 * (a face's bounding sphere, the matrix-core scan's record pieces and record order).  None of them has a counterpart in
 * the reference; a caller that only renders needs rt_hip.h alone.
 */
#ifndef RT_HIP_SELFTEST_H
#define RT_HIP_SELFTEST_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel self-tests used by the parity suite: evaluate the device math on n inputs.
 * fn: 0 sin, 1 cos, 2 acos, 3 pow(x,y), 4 x/y, 5 sqrt(x), 6 next RNG state (state passed
 * in the low 32 bits of x's bit pattern).  x, y, out are HOST double arrays. */
int rt_selftest_math(rt_handle h, int fn, int n, const double* x, const double* y, double* out);

/* Broad-phase self-test: for n independent (ray, bounding sphere) pairs evaluate the FP32
 * conservative test exactly as the render kernel does.  flags[i]: 0 = dropped, 1 = kept for
 * the exact test, 2 = dropped as wholly behind the origin.  Host arrays: origins, dirs,
 * centres are double[n*3], radii double[n]. */
int rt_selftest_broad(rt_handle h, int n, const double* origins, const double* dirs, const double* centres,
                      const double* radii, int32_t* flags);

/* Host-only self-test (no GPU call): the bounding sphere the library gives a face before the FP32 broad phase and the
 * grid may cull it.  Face::intersect (geometry.cpp:160-168) accepts, per vertex, a half-plane bordered by the line through
 * v_i perpendicular to edge i -- the polygon only for rectangles, a larger or unbounded region otherwise -- and the sphere
 * must hold that whole region.  normal: the unit normal as the loader leaves it (readShape.cpp:95).
 * Returns 1 and (centre[3], radius) when the region is bounded, 0 when the face is tested for every ray, < 0 on error. */
int rt_face_bound(int n_verts, const double* verts, const double* normal, double* centre, double* radius);

/* Host-only self-test (no GPU call): how broad_mfma_kernel's record table represents one broad-phase record
 * (cx, cy, cz, w = R^2 - |c|^2) of a scene whose largest record extent is cmax.  scale: the power of two by which lengths
 * are multiplied; pieces[8]: the two f16 pieces (as doubles) of s cx, s cy, s cz and s^2 w.  The pieces of a coordinate
 * sum to it within 2^-21 of its magnitude, those of w to at least s^2 w (w is rounded towards "keep"); an infinite w
 * (never / always kept record) becomes -/+ 65504. */
int rt_mfma_record_pieces(const double* record, double cmax, double* scale, double* pieces);

/* Host-only self-test (no GPU call): the record order of broad_mfma_kernel's table for n bounding spheres (T = 0) and the
 * bounding spheres of its groups of 32.  radii[i] = -1: a solid that is never kept (left out of a reordered table), any
 * other negative radius: an unbounded solid.  reorder = 0: file order.  perm[k] (k < 32 * groups): the index of the solid at
 * place k, 0x7fffffff where a place holds none.  group_spheres[4 g ..]: centre and radius of group g's sphere (radius -1:
 * empty group, +inf: the group holds an unbounded solid).  Returns the number of groups, < 0 on error (too small buffers). */
int rt_mfma_table_order(int n, const double* centres, const double* radii, int reorder, uint32_t* perm, int perm_cap,
                        double* group_spheres, int group_cap);

/* Host-only self-test (no GPU call): the multiplier and shift by which the kernels take n / d for the two divisors of the
 * pixel map (the frame's width, the tile's rows; main.cpp:250-251 walks the same pixels with a counter): for every
 * 0 <= n <= n_max, n / d == (uint32_t)(((uint64_t)n * mul) >> 32) >> shift.  Returns 1 when such a pair exists (d >= 2,
 * n_max < 2^31; the library has checked it at every multiple of d), 0 when the kernels divide instead, < 0 on error. */
int rt_pixel_map_divisor(uint32_t d, uint64_t n_max, uint32_t* mul, uint32_t* shift);

/* Self-test of the broad phase on the matrix cores: every one of n_rays rays against every one of n_rec bounding spheres,
 * evaluated exactly as broad_mfma_kernel does.  masks: uint32[n_rays * ceil(n_rec / 32)] (HOST); bit 31 - k of
 * masks[ray * groups + g] is set iff record 32 g + k is kept for the ray. */
int rt_selftest_broad_mfma(rt_handle h, int n_rays, const double* origins, const double* dirs, int n_rec, const double* centres,
                           const double* radii, uint32_t* masks);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_SELFTEST_H */
