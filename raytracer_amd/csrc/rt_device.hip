// librt_hip.so -- render kernels for gfx950 (MI355X) and their C ABI (include/rt_hip.h; self-tests: include/rt_hip_selftest.h).
//
// What is computed: the per-pixel render loop of samkottler/rayTracer,
// /root/reference/src/main.cpp:247-289 (do_rays_i) with everything it calls:
// add_to_path :126-183, get_intersection :34-43, get_reflection_ray :47-104,
// get_refraction_ray :107-122, the primitives of /root/reference/src/geometry.cpp
// :94-179 and the shading product Path::trace :186-217; and expose() + blur(),
// main.cpp:186-242, on the frame it produces.
//
// How it is organised on the GPU (per-pixel mode, the default):
//   * one RNG stream per pixel (minstd_rand0, the reference's generator, seeded from the pixel's
//     frame index), so a frame does not depend on scheduling or on the multi-GPU partition;
//   * a frame is a sequence of rounds over a queue of live rays; a round is two launches:
//       broad_kernel (FP32, the hot loop): every queued ray against the bounding sphere of every
//         solid, records staged in LDS and read by broadcast, 9 VALU instructions per ray-record
//         pair, branch-free; the error bound of the float evaluation is folded into per-ray and
//         per-record constants, so a solid the exact test could accept is never dropped.  The
//         survivors (a handful per ray) go to the ray's candidate list as 32-record bit masks
//         (walk_kernel, an option, builds the same lists from a walk of a uniform grid);
//       shade_kernel (FP64): the candidates and the always-tested solids (planes, walls) through
//         the reference's own double-precision operation sequence (no fused multiply-add, IEEE
//         divide and sqrt) with get_intersection's acceptance rule and tie order, then one segment
//         of add_to_path; surviving rays are packed into the next round's queue by ballot rank;
//   * the reference's recursion and linked vertex list become the rounds and a per-pixel stack of
//     per-vertex throughput factors, multiplied light-to-camera at a path's end as Path::trace does;
//   * record lists larger than the LDS window are streamed through it window by window.
// render_kernel is the single-kernel form of the same loop (persistent lanes, queue refill by ballot
// rank), used for the reference-order validation mode: one lane per reference thread and RNG stream.
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

#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "../../include/rt_hip_selftest.h"
#include "rt_math.h"

// The translation unit, in dependency order:
#include "rt_tables.h"          // device tables, kernel parameters
#include "rt_geometry.h"        // exact double arithmetic of the reference
#include "rt_broad.h"           // conservative FP32 test
#include "rt_path.h"            // path state, add_to_path segment, narrow phase
#include "rt_kernel_single.h"   // render_kernel
#include "rt_kernel_broad.h"    // WfParams, init_kernel, broad_kernel
#include "rt_kernel_mfma.h"     // broad_mfma_kernel
#include "rt_kernel_walk.h"     // walk_kernel
#include "rt_kernel_shade.h"    // find_hit, shade_kernel
#include "rt_kernel_expose.h"   // expose kernels
#include "rt_context.h"         // rt_context, host helpers

// ---- device memory of rt_expose_device_noalloc: the Gaussian taps and the two buffers the blur passes alternate between,
// for frames of up to RT_NOALLOC_MAX_PIXELS pixels, in a file-scope array that exists on every GPU from the moment the
// library's code object is loaded there.  Layout: [0, NOALLOC_TAPS_BYTES) the 41 x 41 taps (written once per device by
// noalloc_setup), then two buffers of RT_NOALLOC_MAX_PIXELS * 3 doubles.  In host code the array's name is only a
// handle for hipGetSymbolAddress: the address each device holds it at is looked up once per device and kept in
// g_noalloc[device].ptr, and only that pointer is ever passed to kernels or copies.
constexpr size_t NOALLOC_TAPS_BYTES = 16384;                                        // 1681 doubles, padded
constexpr size_t NOALLOC_BUF_BYTES = (size_t)RT_NOALLOC_MAX_PIXELS * 3 * sizeof(double);
constexpr size_t NOALLOC_BYTES = NOALLOC_TAPS_BYTES + 2 * NOALLOC_BUF_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

constexpr int NOALLOC_MAX_DEVICES = 64;
// per device: where that device holds `scratch` (null = not set up) and an event recorded after the last call's launches.
// Every handle and stream on a GPU shares the GPU's array, so a call first makes its stream wait for that event: calls
// that would overlap on one GPU run one after the other instead of racing.  (A stream under graph capture cannot wait for
// an event recorded outside the capture; there the ordering is left to the graph and its caller.)
struct NoallocDevice { unsigned char* ptr = nullptr; hipEvent_t last_use = nullptr; };
static NoallocDevice g_noalloc[NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mutex;     // guards g_noalloc and keeps a call's wait / launches / record together

// once per device, with that device current (rt_create): look up the device's copy of the array and fill in the taps.
// A failure here is not fatal to the context: rt_expose_device_noalloc then refuses, everything else works.
static int noalloc_setup(rt_context* h) {
    if (h->device < 0 || h->device >= NOALLOC_MAX_DEVICES) return RT_OK;            // rt_expose_device_noalloc will refuse
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    NoallocDevice& nd = g_noalloc[h->device];
    if (nd.ptr) return RT_OK;
    unsigned char* ptr = nullptr;
    HIP_TRY(h, hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(h, RT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    const std::vector<double> k = bloom_taps();
    static_assert(BLOOM_K * BLOOM_K * sizeof(double) <= NOALLOC_TAPS_BYTES, "taps do not fit their part of the array");
    HIP_TRY(h, hipMemcpy(ptr, k.data(), k.size() * sizeof(double), hipMemcpyHostToDevice));
    if (!nd.last_use) HIP_TRY(h, hipEventCreateWithFlags(&nd.last_use, hipEventDisableTiming));
    nd.ptr = ptr;
    return RT_OK;
}

// ======================================================================= C ABI (include/rt_hip.h)
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
    if (hipMalloc((void**)&h->d_ctrl, RT_CTRL_BYTES) != hipSuccess) { delete h; return RT_ERR_HIP; }
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        if (h->ev0) hipEventDestroy(h->ev0);
        if (h->ev1) hipEventDestroy(h->ev1);
        hipFree(h->d_ctrl);
        delete h;
        return RT_ERR_HIP;
    }
    if (noalloc_setup(h) != RT_OK) h->err.clear();   // only rt_expose_device_noalloc needs it, and it says so when called
    // tuning knobs for experiments; rt_set_option is the API for the same values
    if (const char* e = getenv("RT_MODE")) rt_set_option(h, RT_OPT_MODE, strcmp(e, "mega") == 0 ? RT_MODE_SINGLE_KERNEL : RT_MODE_PIPELINE);
    if (const char* e = getenv("RT_BLOCK")) rt_set_option(h, RT_OPT_BLOCK, atoi(e));
    if (const char* e = getenv("RT_BROAD_BLOCK")) rt_set_option(h, RT_OPT_BROAD_BLOCK, atoi(e));
    if (const char* e = getenv("RT_WG_PER_CU")) rt_set_option(h, RT_OPT_WG_PER_CU, atoi(e));
    if (const char* e = getenv("RT_CHUNK_RECORDS")) rt_set_option(h, RT_OPT_CHUNK_RECORDS, atoi(e));
    if (const char* e = getenv("RT_OVERLAP")) rt_set_option(h, RT_OPT_OVERLAP, atoi(e));
    if (const char* e = getenv("RT_OVERLAP_MIN_PIXELS")) { const long long v = atoll(e); if (v >= 0) h->overlap_min_pixels = (uint64_t)v; }
    if (const char* e = getenv("RT_ACCEL")) rt_set_option(h, RT_OPT_ACCEL, atoi(e));
    if (const char* e = getenv("RT_ACCEL_MIN_BOUNDED")) h->accel_min_bounded = atoi(e);
    if (const char* e = getenv("RT_NO_BOX")) h->no_box = atoi(e) != 0;
    if (const char* e = getenv("RT_CAND_LDS")) h->cand_lds = atoi(e) != 0;
    if (const char* e = getenv("RT_MFMA_CULL")) h->mf_cull = atoi(e) != 0;
    if (const char* e = getenv("RT_MFMA_BINNED")) h->mf_binned = atoi(e) != 0;
    if (const char* e = getenv("RT_MFMA_BEHIND")) h->mf_behind = atoi(e) != 0;
    if (const char* e = getenv("RT_MFMA_MIN_BOUNDED")) h->mf_min_bounded = atoi(e);
    if (const char* e = getenv("RT_MFMA_BINNED_MIN")) h->mf_binned_min_groups = atoi(e);
    if (const char* e = getenv("RT_MFMA_LIST")) { const int v = atoi(e); if (v >= 0 && v <= MFB_LIST) h->mf_list_cap = v; }
    if (const char* e = getenv("RT_BROAD_MFMA")) h->broad_mfma = atoi(e) != 0;
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
            h->overlap = value; h->overlap_chosen = true; return RT_OK;
        case RT_OPT_ACCEL:
            if (value != RT_ACCEL_SCAN && value != RT_ACCEL_WALK && value != RT_ACCEL_AUTO) return fail(h, RT_ERR_ARG, "rt_set_option: accel must be 0 (scan), 1 (grid walk) or 2 (chosen per scene)");
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

int rt_pixel_map_divisor(uint32_t d, uint64_t n_max, uint32_t* mul, uint32_t* shift) {
    if (!mul || !shift || d == 0) return RT_ERR_ARG;
    unsigned m = 0, s = 0;
    div_magic(d, n_max, m, s);
    *mul = m; *shift = s;
    return m ? 1 : 0;
}

int rt_face_bound(int n_verts, const double* verts, const double* normal, double* centre, double* radius) {
    if (n_verts < 1 || !verts || !normal || !centre || !radius) return RT_ERR_ARG;
    return face_bound(verts, n_verts, normal, centre, radius) ? 1 : 0;
}

int rt_mfma_table_order(int n, const double* centres, const double* radii, int reorder, uint32_t* perm, int perm_cap,
                        double* group_spheres, int group_cap) {
    if (n < 0 || !centres || !radii || !perm || !group_spheres) return RT_ERR_ARG;
    std::vector<float> rec((size_t)(n > 0 ? n : 1) * 4);
    double cmax = 0;
    for (int i = 0; i < n; i++) {
        float* r4 = &rec[4 * (size_t)i];
        if (radii[i] == -1.0) { r4[0] = r4[1] = r4[2] = 0.0f; r4[3] = -INFINITY; continue; }      // never kept (always-tested solid)
        if (radii[i] < 0.0) { r4[0] = r4[1] = r4[2] = 0.0f; r4[3] = INFINITY; continue; }         // unbounded
        const double ext = make_record(centres + 3 * (size_t)i, radii[i], r4);
        if (ext > cmax) cmax = ext;
    }
    const MfTable t = mf_build_table(rec.data(), n, cmax, reorder != 0);
    if ((int)t.perm.size() > perm_cap || t.n_groups > group_cap) return RT_ERR_ARG;
    for (size_t k = 0; k < t.perm.size(); k++) perm[k] = t.perm[k];
    for (int g = 0; g < t.n_groups; g++) {
        const float* gr = &t.group_records[4 * (size_t)g];
        double* o = group_spheres + 4 * (size_t)g;
        o[0] = gr[0]; o[1] = gr[1]; o[2] = gr[2];
        if (!(fabsf(gr[3]) <= 3.0e38f)) o[3] = gr[3] < 0 ? -1.0 : INFINITY;
        else o[3] = sqrt(fmax(0.0, (double)gr[3] + (double)gr[0] * gr[0] + (double)gr[1] * gr[1] + (double)gr[2] * gr[2]));
    }
    return t.n_groups;
}

int rt_mfma_record_pieces(const double* record, double cmax, double* scale, double* pieces) {
    if (!record || !scale || !pieces) return RT_ERR_ARG;
    const float s = mf_scale_for((float)cmax);
    *scale = (double)s;
    for (int i = 0; i < 4; i++) {
        unsigned h, l;
        const float x = (float)record[i] * (i < 3 ? s : s * s);
        if (i == 3 && !(fabsf((float)record[3]) <= 3.0e38f)) { h = record[3] < 0 ? 0xfbffu : 0x7bffu; l = 0u; }   // as mf_record_frag
        else f16_split2(x, i == 3, h, l);
        pieces[2 * i] = (double)f16_value(h);
        pieces[2 * i + 1] = (double)f16_value(l);
    }
    return RT_OK;
}

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
    EventPair tm;
    if (tm.ok()) hipEventRecord(tm.a, 0);
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
        o.inv_sexp = 1.0 / o.sexp;
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
            // the sphere around the region Face::intersect accepts, which is not the polygon (face_bound); a face whose
            // region is unbounded keeps an unbounded sphere (always a candidate)
            double c[3], R = 0;
            if (nv >= 1 && face_bound(verts + (size_t)off * 3, nv, q.n, c, &R)) {
                for (int k = 0; k < 3; k++) bc[(size_t)i * 3 + k] = c[k];
                br[i] = R;
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
    const std::vector<float> rec_file = rec;       // one record per solid in file order, before any padding (mf_build_table indexes it by file index)
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

    // the table of the "always" solids: box entries first (narrow_always), then the others
    std::vector<double> alw;
    int n_always = 0, n_box = 0;
    std::vector<char> is_box((size_t)n_prims + 1, 0);
    for (int i = 0; i < n_prims; i++) {
        if (!always[i] || pd[i].kind == RT_KIND_SPHERE || h->no_box) continue;
        if (!(fabs(pd[i].a[0]) < 1e300 && fabs(pd[i].a[1]) < 1e300 && fabs(pd[i].a[2]) < 1e300)) continue;
        const int nv = pd[i].kind == RT_KIND_FACE ? pd[i].vcnt : 0;
        double blo[3], bhi[3];
        const int code = box_code(pd[i].kind, pd[i].n, vd.data() + (nv ? pd[i].voff : 0), nv, blo, bhi);
        if (!code) continue;
        const int k = (code < 0 ? -code : code) - 1;
        const double ent[ALW_BOX] = {(double)code, (double)i, pd[i].a[k], 0.0, blo[0], blo[1], blo[2], 0.0, bhi[0], bhi[1], bhi[2], 0.0};
        alw.insert(alw.end(), ent, ent + ALW_BOX);
        is_box[i] = 1; n_box++; n_always++;
    }
    for (int i = 0; i < n_prims; i++) {
        if (!always[i] || is_box[i]) continue;
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

    // the matrix-core scan's table; its group records can reach further out than any record, and cmax covers both
    // Its own record order (compact groups of 32, mf_kd_order) lets the scan ask the groups' bounding spheres first (per tile for the camera rays,
    // per ray afterwards); it needs groups enough to drop.  Below that the table keeps the file order (consecutive solids
    // of a scene file tend to be neighbours along some line, which packs a ray's hits into few mask words).
    int n_bounded = 0;
    for (int i = 0; i < n_prims; i++) if (!always[i] && br[i] >= 0) n_bounded++;
    h->n_bounded = n_bounded;
    const bool mf_reorder = h->mf_cull && n_bounded >= h->mf_min_bounded;
    const MfTable mt = mf_build_table(rec_file.data(), n_prims, cmax, mf_reorder);
    if (mt.reordered && mt.cmax > cmax) cmax = mt.cmax;

    GridHost gh;
    const bool grid_ok = build_grid(n_prims, pd, bc, br, always, T, h->grid_density, gh);

    free_scene(h);
    // device tables; a failure part of the way leaves no table behind (have_scene stays false until the end)
    const int rc_tables = [&]() -> int {
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
    h->n_always = n_always; h->always_doubles = always_doubles; h->n_box = n_box;
    HIP_TRY(h, hipMalloc((void**)&h->d_frags, mt.frags.size() * sizeof(uint4)));
    HIP_TRY(h, hipMemcpy(h->d_frags, mt.frags.data(), mt.frags.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (mt.reordered) {
        HIP_TRY(h, hipMalloc((void**)&h->d_mf_perm, mt.perm.size() * sizeof(unsigned)));
        HIP_TRY(h, hipMemcpy(h->d_mf_perm, mt.perm.data(), mt.perm.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        // the exact phase's copy of the solids' first 32 bytes in the table's order
        std::vector<double> mc(mt.perm.size() * 4);
        for (size_t k = 0; k < mt.perm.size(); k++) {
            const unsigned i = mt.perm[k];
            double* c = &mc[4 * k];
            if (i < (unsigned)n_prims) { c[0] = pd[i].a[0]; c[1] = pd[i].a[1]; c[2] = pd[i].a[2]; c[3] = pd[i].radius; }
            else { c[0] = c[1] = c[2] = 0.0; c[3] = NAN; }
        }
        HIP_TRY(h, hipMalloc((void**)&h->d_mf_cand, mc.size() * sizeof(double)));
        HIP_TRY(h, hipMemcpy(h->d_mf_cand, mc.data(), mc.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    h->mf_reordered = mt.reordered;
    h->n_groups = mt.n_groups; h->mf_win_groups = mt.win_groups; h->mf_windows = mt.n_windows; h->mf_scale = mt.scale;
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
        return RT_OK;
    }();
    if (rc_tables != RT_OK) { free_scene(h); return rc_tables; }

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
    h->timing.upload_ms = tm.stop_ms();
    return RT_OK;
}

static int render_single_kernel(rt_handle h, int stream_mode, int tile_rows, int rank, int world, double* d_rgb,
                                int32_t* d_hit_ids, hipStream_t st) {
    KParams kp;
    fill_kparams(h, kp);
    kp.tile_rows = tile_rows; kp.rank = rank; kp.world = world;
    set_pixel_map(kp);
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
    h->accel_used = RT_ACCEL_SCAN;
    h->last_sub = 1;
    kp.rgb = d_rgb; kp.hit_ids = d_hit_ids;
    HIP_TRY(h, hipMemsetAsync(h->d_ctrl, 0, RT_CTRL_BYTES, st));
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
        hipFree(b.stack); hipFree(b.counts); hipFree(b.acc); hipFree(b.cand); hipFree(b.cand_cnt); hipFree(b.tm);
        b = rt_context::PipeBufs();
    }
    HIP_TRY(h, hipMalloc((void**)&b.tm, slots * sizeof(float2)));
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

// does the next pipeline render take its candidate solids from the grid walk?
static bool walk_render(rt_handle h) {
    return h->grid_ok && (h->accel == RT_ACCEL_WALK || (h->accel == RT_ACCEL_AUTO && h->n_bounded > h->accel_min_bounded));
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
    // The split pays from about three trips of shade_kernel's grid per part (1.5 M pixels each): the eighth of a 4K frame
    // that one of 8 ranks renders (1.04 M pixels) takes 0.80 ms in one piece and 0.89 ms in two, a 1920 x 1080 frame 0.47
    // against 0.51 ms; from 4 M pixels on two parts win (half a 4K frame 2.50 against 2.54 ms, an eighth of an 8K frame with
    // the grid walk 5.8 against 6.6 ms).
    // With the grid walk as candidate source three parts do better than two (8K, 4096 spheres: 39.4 against 39.7 ms; 4K, 16 384
    // solids, 16 samples: 268 against 277 ms -- a wave walks as long as its longest ray, which leaves more of a launch's
    // tail to fill); with the scan they are level (8K) or lose (4K: 4.42 against 4.36 ms).  Parts of 1.5 M pixels at least.
    int parts = h->overlap;
    if (!h->overlap_chosen && parts == 2 && walk_render(h) && total >= 3 * (h->overlap_min_pixels / 2)) parts = 3;
    const int nsub = (parts >= 2 && !h->timing_detail && total >= h->overlap_min_pixels && total >= (1u << 16)) ? parts : 1;
    h->last_sub = nsub;
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
        set_pixel_map(w.kp);
        w.kp.max_verts = verts;
        w.kp.total_items = (unsigned)total;
        w.kp.rgb = d_rgb; w.kp.hit_ids = d_hit_ids;
        for (int q = 0; q < 2; q++) { w.si[q] = b.si[q]; w.sd[q] = b.sd[q]; w.queue[q] = b.queue[q]; w.rec[q] = b.rec[q]; }
        w.stack = b.stack; w.counts = b.counts; w.acc = b.acc;
        w.cand = b.cand; w.cand_cnt = b.cand_cnt; w.tm = b.tm;
    }
    unsigned bgrid = 1;
    hipError_t e;
    if (h->broad_block == 1024) e = broad_grid<1024>(h, &bgrid);
    else if (h->broad_block == 256) e = broad_grid<256>(h, &bgrid);
    else e = broad_grid<512>(h, &bgrid);
    if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("broad kernel setup: ") + hipGetErrorString(e));
    // Where a ray's candidate solids come from.  Both sources hand the exact phase a superset of what it can accept, so the
    // frame is the same bit for bit; left to the library the choice follows the measured crossover (4K room, ms per frame scan /
    // walk): 1024 solids 4.33 / 5.28, 1280: 5.18 / 5.57, 1792: 5.51 / 6.02, 2048: 5.31 / 6.17, 2560: 6.36 / 6.58, 3072: 6.43 / 6.82,
    // 4096: 7.50 / 6.79, 6144: 8.48 / 7.67 -- the scan up to three LDS windows of the binned kernel (3072 bounded solids), the
    // walk beyond
    const bool use_grid = walk_render(h);
    h->accel_used = use_grid ? RT_ACCEL_WALK : RT_ACCEL_SCAN;
    // the scan on the matrix cores (default) or on the vector ALU; per-launch workgroup size 512 either way
    const bool use_mfma = h->broad_mfma && !use_grid && h->d_frags;
    // one LDS window: its groups' fragments and those of its group records
    const int mf_win_groups = h->mf_win_groups;
    const size_t mfma_lds = (size_t)(mf_win_groups + (mf_win_groups + BROAD_GROUP - 1) / BROAD_GROUP) * MF_GROUP_BYTES;
    // the binned scan: 1024 threads, two workgroups per CU (LDS: the window, 512 rays' fragments, the groups' ray lists)
    constexpr int MFB_BLOCK = 1024, MFB_RAYS = MFB_BLOCK / 2;
    const size_t binned_lds = (size_t)(mf_win_groups + 1) * MF_GROUP_BYTES + (size_t)(MFB_RAYS + 1) * 64 + (size_t)MFB_GROUPS * MFB_LIST * 2 +
                              (size_t)MFB_RAYS * 4 + 2 * MFB_GROUPS * 4;
    const bool use_binned = h->mf_reordered && h->mf_binned && h->n_groups >= h->mf_binned_min_groups;
    unsigned mgrid = 1, mgrid_cull = 1;
    if (use_mfma) {
        e = mfma_grid<MF_BLOCK, false>(h, mfma_lds, &mgrid);
        if (e == hipSuccess && h->mf_reordered) e = mfma_grid<MF_BLOCK, true>(h, mfma_lds, &mgrid_cull);
        if (e == hipSuccess && use_binned)
            e = hipFuncSetAttribute((const void*)broad_mfma_binned_kernel<MFB_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)binned_lds);
        if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("broad_mfma_kernel setup: ") + hipGetErrorString(e));
        for (int p = 0; p < nsub; p++) { wp[p].kp.wide = 1; wp[p].kp.mf_perm = h->d_mf_perm; wp[p].kp.mf_cand = h->d_mf_cand; }
    }
    // shade_kernel: 8 workgroups per CU, two generations of the 4 that are resident (C3 frame: 4 -> 6.17, 8 -> 6.06,
    // 16 -> 6.15, 32 -> 6.27 ms)
    unsigned sgrid = (unsigned)h->cu_count * 8u;
    if (const char* e = getenv("RT_SHADE_GRID")) sgrid = (unsigned)h->cu_count * (unsigned)atoi(e);
    h->last_streams = 0;
    h->broad_ev_used = 0;
    if (h->timing_detail) {
        const size_t nb = (total + slots_cap[0] - 1) / slots_cap[0];
        const size_t need = (size_t)3 * iters * (nb ? nb : 1);       // per round: before the scan, between scan and shade_kernel, after
        while (h->broad_ev.size() < need) { hipEvent_t ev; HIP_TRY(h, hipEventCreate(&ev)); h->broad_ev.push_back(ev); }
    }
    HIP_TRY(h, hipMemsetAsync(h->d_ctrl, 0, RT_CTRL_BYTES, st));
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
    // shade_kernel's LDS: the always table and, when the scan's table is small enough for four workgroups per CU to hold a
    // copy each (1024 places: 36 KB), the exact phase's compact records of it -- a candidate's record then comes out of
    // LDS in a tenth of the time it takes out of the L2
    size_t shade_lds = (size_t)((h->always_doubles + 1) & ~1) * sizeof(double);
    int places_lds = 0;
    if (use_mfma && h->mf_reordered && h->cand_lds) {
        const size_t extra = (size_t)h->n_groups * BROAD_GROUP * (4 * sizeof(double) + sizeof(unsigned));
        if (shade_lds + extra <= (size_t)40 * 1024) {          // a quarter of a CU's 160 KB: the kernel's four workgroups per CU stay
            places_lds = h->n_groups * BROAD_GROUP;
            shade_lds += extra;
        }
    }
    for (int p = 0; p < nsub; p++) wp[p].kp.mf_places_lds = places_lds;
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
            w.iter = 0;
            unsigned igrid = (unsigned)((n[p] + 255) / 256);
            if (igrid > (unsigned)h->cu_count * 16u) igrid = (unsigned)h->cu_count * 16u;
            hipLaunchKernelGGL(init_kernel, dim3(igrid), dim3(256), 0, ss[p], w, (unsigned)iters + 2u);
        }
        if (!any) break;
        for (int it = 0; it < iters; it++) {
            for (int p = 0; p < nsub; p++) {
                if (!n[p]) continue;
                WfParams& w = wp[p];
                w.iter = it;
                if (h->timing_detail) HIP_TRY(h, hipEventRecord(h->broad_ev[h->broad_ev_used++], ss[p]));
                if (use_grid) { hipLaunchKernelGGL(walk_kernel, dim3((unsigned)h->cu_count * 16u), dim3(256), 0, ss[p], w, h->grid); e = hipGetLastError(); }
                else if (use_mfma) {
                    // the camera rays' round of a table in its own order asks the group records first
                    if (it == 0 && h->mf_reordered)
                        hipLaunchKernelGGL((broad_mfma_kernel<MF_BLOCK, true>), dim3(mgrid_cull), dim3(MF_BLOCK), mfma_lds, ss[p], w, (const uint4*)h->d_frags, h->n_groups, mf_win_groups, h->mf_scale);
                    else if (use_binned)
                        hipLaunchKernelGGL(broad_mfma_binned_kernel<MFB_BLOCK>, dim3((unsigned)h->cu_count * 2u), dim3(MFB_BLOCK), binned_lds, ss[p], w, (const uint4*)h->d_frags, h->n_groups, mf_win_groups, h->mf_scale, h->mf_list_cap, h->mf_behind ? 1 : 0);
                    else
                        hipLaunchKernelGGL((broad_mfma_kernel<MF_BLOCK, false>), dim3(mgrid), dim3(MF_BLOCK), mfma_lds, ss[p], w, (const uint4*)h->d_frags, h->n_groups, mf_win_groups, h->mf_scale);
                    e = hipGetLastError();
                }
                else if (h->broad_block == 1024) e = launch_broad<1024>(h, w, bgrid, ss[p]);
                else if (h->broad_block == 256) e = launch_broad<256>(h, w, bgrid, ss[p]);
                else e = launch_broad<512>(h, w, bgrid, ss[p]);
                if (e != hipSuccess) return fail(h, RT_ERR_HIP, std::string("broad kernel launch: ") + hipGetErrorString(e));
                if (h->timing_detail) HIP_TRY(h, hipEventRecord(h->broad_ev[h->broad_ev_used++], ss[p]));
                const dim3 sg(sgrid * 256 / SHADE_BLOCK), sb(SHADE_BLOCK);
                if (w.kp.mf_perm) {
                    if (it == 0) hipLaunchKernelGGL((shade_kernel<true, true>), sg, sb, shade_lds, ss[p], w);
                    else hipLaunchKernelGGL((shade_kernel<false, true>), sg, sb, shade_lds, ss[p], w);
                } else {
                    if (it == 0) hipLaunchKernelGGL((shade_kernel<true, false>), sg, sb, shade_lds, ss[p], w);
                    else hipLaunchKernelGGL((shade_kernel<false, false>), sg, sb, shade_lds, ss[p], w);
                }
                if (h->timing_detail) HIP_TRY(h, hipEventRecord(h->broad_ev[h->broad_ev_used++], ss[p]));
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

}  // extern "C"

// what the guard word of a finished render says; ctrl: the first 8 words of the context's control block
static int guard_status(rt_handle h, const unsigned* ctrl) {
    if (ctrl[1] & ERR_REJECT_GUARD)
        return fail(h, RT_ERR_GUARD, "render: the glossy rejection loop did not accept within its guard at local pixel " + std::to_string(ctrl[7] - 1u) +
                                     " (a glossy surface hit from behind: the reference never terminates on this scene)");
    if (ctrl[1] & ERR_STACK) return fail(h, RT_ERR_GUARD, "render: bounce stack overflow (path longer than 2*max_ray_depth+2 vertices: no emitter in the scene?)");
    return RT_OK;
}

extern "C" {

int rt_finish(rt_handle h, uint64_t* rays) {
    if (!h) return RT_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->events_pending) return fail(h, RT_ERR_ARG, "rt_finish: no render in flight");
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    h->events_pending = false;
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    unsigned ctrl[16];
    HIP_TRY(h, hipMemcpy(ctrl, h->d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
    uint64_t r = (uint64_t)ctrl[2] | ((uint64_t)ctrl[3] << 32);
    h->timing.kernel_ms = ms;
    h->timing.broad_ms = 0.0;
    h->timing.shade_ms = 0.0;
    for (int i = 0; i + 2 < h->broad_ev_used; i += 3) {
        float bms = 0, sms = 0;
        if (hipEventElapsedTime(&bms, h->broad_ev[i], h->broad_ev[i + 1]) == hipSuccess) h->timing.broad_ms += bms;
        if (hipEventElapsedTime(&sms, h->broad_ev[i + 1], h->broad_ev[i + 2]) == hipSuccess) h->timing.shade_ms += sms;
    }
    h->timing.rays = r;
    h->timing.candidates = (uint64_t)ctrl[4] | ((uint64_t)ctrl[5] << 32);
    h->timing.overflow_rays = ctrl[6];
    h->timing.broad_tiles = (uint64_t)ctrl[14] | ((uint64_t)ctrl[15] << 32);
    h->timing.broad_records = (uint64_t)h->n_groups * BROAD_GROUP;
    h->timing.accel_used = (uint64_t)h->accel_used;
    h->timing.sub_pipelines = (uint64_t)h->last_sub;
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
        unsigned rc[2];
        hipMemcpy(rc, h->d_ctrl + 30, sizeof(rc), hipMemcpyDeviceToHost);
        {   // lanes active where a section of the kernel is entered (PROF_LANES points), summed over the frame's launches
            unsigned long long la[2 * 24];
            hipMemcpy(la, h->d_ctrl + 64, sizeof(la), hipMemcpyDeviceToHost);
            static const char* names[24] = {"candidate test", "always tests", "segment", "sphere normal", "plane material", "factor", "factor pow",
                                            "light product step", "dead end", "reflect", "refract", "end_path", "emit", "heavy ray", "retry pass", "state load",
                                            "list scan step", "walk: ray", "walk: step", "walk: first item", "walk: further item", "", "", ""};
            for (int k = 0; k < 24; k++)
                if (la[2 * k + 1]) fprintf(stderr, "  lanes %-20s entries %10llu  mean active %5.1f of 64\n", names[k], la[2 * k + 1], (double)la[2 * k] / (double)la[2 * k + 1]);
        }
        fprintf(stderr, "shade wave-cycles: loop-top %llu  find_hit %llu  segment %llu  retry pool %llu  barriers %llu  emit %llu   (waves with a rejected sample %u, such rays %u)\n",
                prof[0], prof[1], prof[2], prof[5], prof[3], prof[4], rc[0], rc[1]);
    }
#endif
    h->timing.sphere_tests = r * h->n_spheres;
    h->timing.plane_tests = r * h->n_planes;
    h->timing.face_tests = r * h->n_faces;
    h->timing.face_vert_tests = r * h->n_face_verts;
    if (rays) *rays = r;
    return guard_status(h, ctrl);
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
    EventPair tm;
    if (tm.ok()) hipEventRecord(tm.a, 0);
    const size_t npx = (size_t)h->width * h->height;
    const size_t rb = npx * 3 * sizeof(double), hb = npx * h->spp * sizeof(int32_t);
    if (h->d_rgb_bytes < rb) { hipFree(h->d_rgb); h->d_rgb = nullptr; h->d_rgb_bytes = 0; HIP_TRY(h, hipMalloc((void**)&h->d_rgb, rb)); h->d_rgb_bytes = rb; }
    if (hit_ids && h->d_hits_bytes < hb) { hipFree(h->d_hits); h->d_hits = nullptr; h->d_hits_bytes = 0; HIP_TRY(h, hipMalloc((void**)&h->d_hits, hb)); h->d_hits_bytes = hb; }
    int rc = render_common(h, stream_mode, h->height, 0, 1, h->d_rgb, hit_ids ? h->d_hits : nullptr, nullptr);
    if (rc != RT_OK) return rc;
    rc = rt_finish(h, rays);
    HIP_TRY(h, hipMemcpy(rgb, h->d_rgb, rb, hipMemcpyDeviceToHost));
    if (hit_ids) HIP_TRY(h, hipMemcpy(hit_ids, h->d_hits, hb, hipMemcpyDeviceToHost));
    h->timing.total_ms = tm.stop_ms();
    return rc;
}

int rt_render(rt_handle h, double* rgb, uint64_t* rays, int32_t* hit_ids) { return render_host(h, 0, rgb, rays, hit_ids); }

int rt_render_streams(rt_handle h, double* rgb, uint64_t* rays, int32_t* hit_ids) { return render_host(h, 1, rgb, rays, hit_ids); }

// the Gaussian taps and the two frame-sized buffers the blur passes alternate between (n pixels)
static int ensure_bloom(rt_handle h, size_t n) {
    if (!h->d_bloom_kern) {
        const std::vector<double> k = bloom_taps();
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

// rt_expose_device without a device allocation: the blur passes alternate between the two buffers of the file-scope
// array (through the pointer noalloc_setup looked up for this device), the taps are the ones stored at its start.
// Nothing of the array has to start at zero: expose_mask_kernel writes every element of B0 that the frame has and every
// blur pass writes every element of its output before the next pass or the pack reads it.
int rt_expose_device_noalloc(rt_handle h, const double* d_rgb, int width, int height, double exposure, int num_blurs,
                             int32_t* d_img, void* stream) {
    if (!h) return RT_ERR_ARG;
    if (!d_rgb || !d_img || width < 1 || height < 1 || num_blurs < 0) return fail(h, RT_ERR_ARG, "rt_expose_device_noalloc: bad argument");
    const size_t n = (size_t)width * (size_t)height;
    if (n > (size_t)RT_NOALLOC_MAX_PIXELS)
        return fail(h, RT_ERR_ARG, "rt_expose_device_noalloc: frame of " + std::to_string(n) + " pixels exceeds RT_NOALLOC_MAX_PIXELS (" +
                                   std::to_string(RT_NOALLOC_MAX_PIXELS) + "); use rt_expose_device");
    if (h->device < 0 || h->device >= NOALLOC_MAX_DEVICES) return fail(h, RT_ERR_HIP, "rt_expose_device_noalloc: device index beyond the library's table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    const NoallocDevice& nd = g_noalloc[h->device];
    unsigned char* const base = nd.ptr;
    if (!base) return fail(h, RT_ERR_HIP, "rt_expose_device_noalloc: the static buffer of this device was not set up by rt_create");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    // wait for the previous call's use of the array (any stream, any handle on this GPU); not under graph capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool ordered = cap == hipStreamCaptureStatusNone && nd.last_use != nullptr;
    if (ordered && num_blurs > 0) HIP_TRY(h, hipStreamWaitEvent(st, nd.last_use, 0));
    const double* taps = (const double*)base;
    double* buf[2] = {(double*)(base + NOALLOC_TAPS_BYTES), (double*)(base + NOALLOC_TAPS_BYTES + NOALLOC_BUF_BYTES)};
    unsigned grid = (unsigned)((n + 255) / 256);
    if (grid > (unsigned)h->cu_count * 32u) grid = (unsigned)h->cu_count * 32u;
    if (num_blurs == 0) {
        hipLaunchKernelGGL(expose_pack_kernel, dim3(grid), dim3(256), 0, st, d_rgb, (const double*)nullptr, d_img, n, exposure);
        HIP_TRY(h, hipGetLastError());
        return RT_OK;
    }
    unsigned g3 = (unsigned)((n * 3 + 255) / 256);
    if (g3 > (unsigned)h->cu_count * 32u) g3 = (unsigned)h->cu_count * 32u;
    hipLaunchKernelGGL(expose_mask_kernel, dim3(g3), dim3(256), 0, st, d_rgb, buf[0], n * 3, exposure);
    int cur = 0;
    const dim3 bgrid((width + BLOOM_T - 1) / BLOOM_T, (height + BLOOM_T - 1) / BLOOM_T, 3);
    for (int b = 0; b < num_blurs; b++) {
        hipLaunchKernelGGL(bloom_kernel, bgrid, dim3(BLOOM_T * BLOOM_T), 0, st, (const double*)buf[cur], buf[cur ^ 1], width, height, taps,
                           (const int*)nullptr);
        cur ^= 1;
    }
    hipLaunchKernelGGL(expose_pack_kernel, dim3(grid), dim3(256), 0, st, d_rgb, (const double*)buf[cur], d_img, n, exposure);
    HIP_TRY(h, hipGetLastError());
    if (ordered) HIP_TRY(h, hipEventRecord(nd.last_use, st));
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
    DevMem mx, my, mout;
    size_t b = (size_t)n * sizeof(double);
    HIP_TRY(h, mx.alloc(b));
    HIP_TRY(h, my.alloc(b));
    HIP_TRY(h, mout.alloc(b));
    double *dx = mx.as<double>(), *dy = my.as<double>(), *dout = mout.as<double>();
    HIP_TRY(h, hipMemcpy(dx, x, b, hipMemcpyHostToDevice));
    if (y) HIP_TRY(h, hipMemcpy(dy, y, b, hipMemcpyHostToDevice));
    else HIP_TRY(h, hipMemset(dy, 0, b));
    hipLaunchKernelGGL(selftest_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, fn, n, dx, dy, dout);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(out, dout, b, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_selftest_broad_mfma(rt_handle h, int n_rays, const double* origins, const double* dirs, int n_rec, const double* centres,
                           const double* radii, uint32_t* masks) {
    if (!h || n_rays < 0 || n_rec < 0 || !origins || !dirs || !centres || !radii || !masks) return RT_ERR_ARG;
    if (n_rays == 0 || n_rec == 0) return RT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // records as rt_upload_scene makes them (T = 0, cmax over the batch), padded to whole groups with never-kept records
    std::vector<float> rec((size_t)n_rec * 4);
    double cmax = 0;
    for (int i = 0; i < n_rec; i++) {
        const double ext = make_record(centres + 3 * (size_t)i, fabs(radii[i]), &rec[4 * (size_t)i]);
        if (ext > cmax) cmax = ext;
    }
    while ((rec.size() / 4) % BROAD_GROUP) { rec.push_back(0.0f); rec.push_back(0.0f); rec.push_back(0.0f); rec.push_back(-INFINITY); }
    const float scale = mf_scale_for((float)(cmax * (1.0 + 1e-6)));
    const std::vector<uint4> fr = mf_build_frags(rec, scale);
    const int n_groups = (int)(fr.size() / MF_GROUP_WORDS);
    DevMem mO, mD, mA, mB, mF, mM;
    HIP_TRY(h, mO.alloc((size_t)n_rays * 24));
    HIP_TRY(h, mD.alloc((size_t)n_rays * 24));
    HIP_TRY(h, mA.alloc((size_t)n_rays * 16));
    HIP_TRY(h, mB.alloc((size_t)n_rays * 16));
    HIP_TRY(h, mF.alloc(fr.size() * sizeof(uint4)));
    HIP_TRY(h, mM.alloc((size_t)n_rays * n_groups * 4));
    HIP_TRY(h, hipMemcpy(mO.p, origins, (size_t)n_rays * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(mD.p, dirs, (size_t)n_rays * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(mF.p, fr.data(), fr.size() * sizeof(uint4), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_rayrec_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, 0, n_rays, mO.as<double>(), mD.as<double>(),
                       (float)(cmax * (1.0 + 1e-6)), 1, mA.as<float4>(), mB.as<float4>());
    unsigned grid = (unsigned)((n_rays + 31) / 32);
    if (grid > 4096u) grid = 4096u;
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(grid), dim3(64), 0, 0, n_rays, mA.as<float4>(), mB.as<float4>(), mF.as<uint4>(), n_groups,
                       scale, mM.as<unsigned>());
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(masks, mM.p, (size_t)n_rays * n_groups * 4, hipMemcpyDeviceToHost));
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
    DevMem mO, mD, mR, mF;
    HIP_TRY(h, mO.alloc((size_t)n * 24));
    HIP_TRY(h, mD.alloc((size_t)n * 24));
    HIP_TRY(h, mR.alloc((size_t)n * 16));
    HIP_TRY(h, mF.alloc((size_t)n * 4));
    double *dO = mO.as<double>(), *dD = mD.as<double>(); float4* dR = mR.as<float4>(); int* dF = mF.as<int>();
    HIP_TRY(h, hipMemcpy(dO, origins, (size_t)n * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dD, dirs, (size_t)n * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dR, rec.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_broad_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, dO, dD, dR,
                       (float)(cmax * (1.0 + 1e-6)), 0.0, 0.0, 0.0, dF);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(flags, dF, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"

#include "rt_multi.h"           // rt_multi_*: several GPUs in one process, RCCL gather
