// Device-side scene tables and kernel parameters.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

// ------------------------------------------------------------------ device data
constexpr double RT_EPS = 0.001;               // Sphere/Plane::intersect, geometry.cpp:102,135-136
constexpr int RT_MAXV = 2 * RT_MAX_DEPTH + 2;  // bounce-stack entries per lane
constexpr int BROAD_UNROLL = 4;
constexpr int CAND_CAP = 48;                   // candidate list entries per lane before an early narrow phase
constexpr unsigned ERR_REJECT_GUARD = 1u, ERR_STACK = 2u;
constexpr int REJECT_GUARD_ITERS = 1 << 20;
constexpr float F32_U = 5.9604645e-8f;         // 2^-24, unit roundoff of float

constexpr int RT_CTRL_BYTES = 1024;           // the control block: words 0..63 as listed at KParams::ctrl, 64.. profile counters
// SHADE_PROFILE builds: lanes active at point k of shade_kernel (sum, entries) as two u64 at ctrl + 64 + 4k
#ifndef SHADE_PROFILE
#define SHADE_PROFILE 0
#endif
#if SHADE_PROFILE
#define PROF_LANES(ctrl, k) { const unsigned long long pm_ = __ballot(1); if ((int)(threadIdx.x & 63) == __builtin_ctzll(pm_)) { \
    atomicAdd((unsigned long long*)((ctrl) + 64 + 4 * (k)), (unsigned long long)__popcll(pm_)); atomicAdd((unsigned long long*)((ctrl) + 66 + 4 * (k)), 1ull); } }
#else
#define PROF_LANES(ctrl, k)
#endif

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

struct alignas(16) MatD {    // struct Material (geometry.hpp:18-29) in double, 160 bytes
    unsigned is_light;
    unsigned gloss_thr;      // glossy iff rng < gloss_thr  (= ceil(gloss * 2147483646), main.cpp:48)
    double rindex;           // the three above: what a hit decides on, one 16-byte load
    double sexp;
    double inv_sexp;         // 1 / specular_exp (main.cpp:62), divided once on the host: the same IEEE operation
    double ref[3];
    double pad;
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
    // quotients by width and by tile_rows as a multiply-high and a shift (set_pixel_map on the host; mul = 0: divide)
    unsigned div_w_mul, div_w_sh, div_t_mul, div_t_sh;
    int stream_mode, n_streams;
    unsigned total_items;
    int nchunks;
    int n_prims;
    int max_verts;           // bounce-stack entries available per path (<= RT_MAXV)
    float cmax;              // max |C - T| + R over finite bounding spheres
    int wide;                // ray records with the wide error bounds (the broad phase runs on the matrix cores)
    const unsigned* mf_perm; // candidate lists name places of broad_mfma_kernel's record order: place -> file index (null: lists
                             // name file indices)
    int mf_places_lds;       // places of mf_cand / mf_perm that shade_kernel copies into LDS behind the always table (0: none)
    const double* mf_cand;   // [places][4]: (point a, radius) of the solid at a place, NaN radius for a plane, a face or no solid
    double T[3];             // translation applied before rounding coordinates to float
    const float4* records;
    const ChunkDesc* chunks;
    const PrimD* prims;
    const FaceV* verts;
    const MatD* mats;
    const double* always;    // solids every ray must test exactly (unbounded or very large bounding sphere): per solid
                             // layout at narrow_always; their broad-phase records are padding
    int n_always, always_doubles;
    int n_box;               // the first n_box entries of `always` are box entries (narrow_always)
    double* rgb;
    int* hit_ids;
    unsigned long long* stream_rays;
    unsigned* ctrl;          // [0] queue head, [1] error bits, [2..3] rays (u64), [4..5] candidates (u64), [6] rays whose candidate list overflowed, [14..15] (tile, group) products of broad_mfma_kernel (u64)
};

}  // namespace
