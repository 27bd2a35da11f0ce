// Host side: the context behind rt_handle, grid construction, kernel launch helpers.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

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
    uint4* d_frags = nullptr; int n_groups = 0;   // broad_mfma_kernel: record fragments, 1 KiB per group of 32 records
    float mf_scale = 1.0f;                        //   and the power of two by which its lengths are scaled
    int mf_win_groups = 0, mf_windows = 0;        //   groups per LDS window, windows
    bool mf_cull = true;                          //   camera rays' round: group records first (RT_MFMA_CULL=0: scan every group)
    bool mf_reordered = false;                    //   the table is in its own order, compact groups (else file order, and no culling)
    bool mf_binned = true;                        //   rounds after the camera rays': the binned scan (RT_MFMA_BINNED=0: plain)
    bool mf_behind = true;                        //   binned scan: groups wholly behind a ray's origin are not listed (RT_MFMA_BEHIND=0: listed)
    // measured on the 4K room with 256 .. 2048 spheres (tools/lattice_probe.py): file order wins up to 256 solids, own order
    // + group records per tile in round 0 from 512, the binned scan in the later rounds from about 1000 (28 groups)
    int mf_min_bounded = 16 * BROAD_GROUP;        //   bounded solids from which the table takes its own order (RT_MFMA_MIN_BOUNDED)
    int mf_binned_min_groups = 28;                //   groups from which the later rounds use the binned scan (RT_MFMA_BINNED_MIN)
    int mf_list_cap = MFB_LIST;                   //   places used in a group's ray list (RT_MFMA_LIST: tests of the all-rays path)
    unsigned* d_mf_perm = nullptr;                //   place in the table's order -> file index of the solid
    double* d_mf_cand = nullptr;                  //   place -> (point a, radius) of the solid, for the exact phase
    void* d_always = nullptr; int n_always = 0, always_doubles = 0, n_box = 0;
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
        float2* tm = nullptr;
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
    bool overlap_chosen = false;   // set by rt_set_option / RT_OVERLAP: the library then does not pick three parts for a grid-walk render
    uint64_t overlap_min_pixels = 3ull << 20;   // ... for renders of at least this many pixels (RT_OVERLAP_MIN_PIXELS), see render_pipeline
    int last_sub = 1;          // sub-pipelines of the last render
    int broad_mfma = 1;        // 1: the broad phase runs on the matrix cores (broad_mfma_kernel), 0: on the vector ALU (broad_kernel)
    bool cand_lds = true;      // shade_kernel keeps the scan table's compact records in LDS when they fit (RT_CAND_LDS=0: reads them from memory)
    bool no_box = false;       // diagnostic: always-tested planes and faces through the general test only
    int accel = RT_ACCEL_AUTO; // pipeline: candidate lists from the scan of every record (0), from the grid walk (1), or whichever is
                               // faster for the scene (2: the walk above accel_min_bounded bounded solids)
    int accel_min_bounded = 3 * MFB_GROUPS * BROAD_GROUP;   // RT_ACCEL_MIN_BOUNDED: the walk is chosen when the scene has more bounded solids (three LDS windows of the binned scan)
                                                        // outside the always-tested table than this (1024: one window of the binned scan)
    int n_bounded = 0;                   // such solids in the uploaded scene
    int accel_used = RT_ACCEL_SCAN;      // what the last render used
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

// a pair of timing events that every return path destroys; ok() is false when they could not be created
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() { if (hipEventCreate(&a) != hipSuccess) a = nullptr; if (hipEventCreate(&b) != hipSuccess) b = nullptr; }
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
    bool ok() const { return a && b; }
    // milliseconds from a to b, both recorded on the null stream; 0 when anything fails
    double stop_ms() {
        float ms = 0;
        if (!ok() || hipEventRecord(b, 0) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0;
        return ms;
    }
};

// device memory that every return path frees
struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) hipFree(p); }
    DevMem() = default;
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return (T*)p; }
};

void free_scene(rt_context* h) {
    hipFree(h->d_records); hipFree(h->d_chunks); hipFree(h->d_prims); hipFree(h->d_verts); hipFree(h->d_mats); hipFree(h->d_always);
    hipFree(h->d_cells); hipFree(h->d_item_rec); hipFree(h->d_item_aux); hipFree(h->d_frags); hipFree(h->d_mf_perm); hipFree(h->d_mf_cand);
    h->d_frags = nullptr; h->d_mf_perm = nullptr; h->d_mf_cand = nullptr; h->n_groups = 0;
    h->d_cells = nullptr; h->d_item_rec = nullptr; h->d_item_aux = nullptr; h->grid_ok = false;
    h->d_records = h->d_chunks = h->d_prims = h->d_verts = h->d_mats = h->d_always = nullptr;
    h->have_scene = false;
}

void free_pipeline(rt_context* h) {
    for (auto& b : h->pb) {
        for (int q = 0; q < 2; q++) { hipFree(b.si[q]); hipFree(b.sd[q]); hipFree(b.queue[q]); hipFree(b.rec[q]); }
        hipFree(b.stack); hipFree(b.counts); hipFree(b.acc); hipFree(b.cand); hipFree(b.cand_cnt); hipFree(b.tm);
        b = rt_context::PipeBufs();
    }
}

// the Gaussian of main.cpp:188-194, evaluated on the host with the C library's exp like the reference
std::vector<double> bloom_taps() {
    const double stddev = 3, denom = 2 * stddev * stddev;
    std::vector<double> k((size_t)BLOOM_K * BLOOM_K);
    for (int y = -BLOOM_R; y <= BLOOM_R; y++)
        for (int x = -BLOOM_R; x <= BLOOM_R; x++)
            k[(size_t)(y + BLOOM_R) * BLOOM_K + x + BLOOM_R] = exp(-(x * x + y * y) / denom) / denom / M_PI;
    return k;
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


// Bounding sphere of the points Face::intersect can return (geometry.cpp:160-168).  The reference does NOT test the
// polygon: a point p of the plane (verts[0], n) is accepted iff for every i the products
//     to_p * to_next,   to_p = e_i.(v_i - p),   to_next = e_i.(v_i - v_i+2),   e_i = v_i - v_i+1
// are not negative -- per vertex one half-plane whose border passes through v_i PERPENDICULAR to edge i, on the side
// of v_i+2's projection onto that edge; a vertex whose to_next is zero constrains nothing.  The accepted region is the
// intersection of those half-planes: the polygon itself for a rectangle, a larger convex region for any other shape
// (an acute triangle: a larger triangle around it; a right triangle or a parallelogram: an unbounded wedge or strip).
// The region is built by clipping a square of half-width 1e4 x (extent of the vertices) in the face's plane with every
// half-plane, each moved outwards by a tolerance; constraints with a to_next that rounding could flip are left out
// (both only enlarge the region).  Returns false (unbounded: the face is tested for every ray) when the clipped
// region still touches the square, is empty, or the input is not finite; else centre and radius of a sphere around it.
bool face_bound(const double* v, int nv, const double* n, double* centre, double* radius) {
    if (nv < 1) return false;
    for (int k = 0; k < 3 * nv; k++) if (!std::isfinite(v[k])) return false;
    const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!std::isfinite(nl) || !(fabs(nl - 1.0) < 1e-6)) return false;
    double cen[3] = {0, 0, 0};
    for (int k = 0; k < nv; k++) for (int a = 0; a < 3; a++) cen[a] += v[3 * k + a] / nv;
    double ext = 0;
    for (int k = 0; k < nv; k++) {
        double dd = 0;
        for (int a = 0; a < 3; a++) dd += (v[3 * k + a] - cen[a]) * (v[3 * k + a] - cen[a]);
        ext = fmax(ext, sqrt(dd));
    }
    if (!(ext > 1e-12) || !std::isfinite(ext)) return false;   // (a smaller face: products below could underflow)
    // frame of the plane: origin = the vertex centroid dropped onto the plane through verts[0], axes u, w
    double off = 0;
    for (int a = 0; a < 3; a++) off += (cen[a] - v[a]) * n[a];
    double org[3];
    for (int a = 0; a < 3; a++) org[a] = cen[a] - off * n[a];
    int least = 0;
    for (int a = 1; a < 3; a++) if (fabs(n[a]) < fabs(n[least])) least = a;
    double ax[3] = {0, 0, 0}; ax[least] = 1.0;
    double u[3] = {n[1] * ax[2] - n[2] * ax[1], n[2] * ax[0] - n[0] * ax[2], n[0] * ax[1] - n[1] * ax[0]};
    const double ul = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int a = 0; a < 3; a++) u[a] /= ul;
    const double w[3] = {n[1] * u[2] - n[2] * u[1], n[2] * u[0] - n[0] * u[2], n[0] * u[1] - n[1] * u[0]};
    const double L = 1e4 * (ext + fabs(off));
    std::vector<double> poly = {-L, -L, L, -L, L, L, -L, L}, next;
    for (int i = 0; i < nv; i++) {
        const double* vi = v + 3 * i; const double* v1 = v + 3 * ((i + 1) % nv); const double* v2 = v + 3 * ((i + 2) % nv);
        double e[3], to_next = 0, el = 0, wl = 0, k0 = 0, ku = 0, kw = 0;
        for (int a = 0; a < 3; a++) {
            e[a] = vi[a] - v1[a];
            const double t = vi[a] - v2[a];
            to_next += e[a] * t; el += e[a] * e[a]; wl += t * t;
            k0 += e[a] * (vi[a] - org[a]); ku += e[a] * u[a]; kw += e[a] * w[a];
        }
        el = sqrt(el); wl = sqrt(wl);
        if (!(fabs(to_next) > 1e-9 * el * wl) || el < 1e-100 || fabs(to_next) < 1e-200) continue;   // zero, or a sign rounding could decide
                                                                     // (or to_p * to_next could underflow to 0): no constraint
        if (!(fabs(ku) > 1e-12 * el || fabs(kw) > 1e-12 * el)) continue;   // edge perpendicular to the plane: all or nothing
        const double sg = to_next > 0 ? 1.0 : -1.0, tol = 1e-9 * el * (ext + fabs(off));
        // keep g(a, b) = sg (k0 - a ku - b kw) + tol >= 0
        next.clear();
        const size_t m = poly.size() / 2;
        for (size_t j = 0; j < m; j++) {
            const double a0 = poly[2 * j], b0 = poly[2 * j + 1], a1 = poly[2 * ((j + 1) % m)], b1 = poly[2 * ((j + 1) % m) + 1];
            const double g0 = sg * (k0 - a0 * ku - b0 * kw) + tol, g1 = sg * (k0 - a1 * ku - b1 * kw) + tol;
            if (g0 >= 0) { next.push_back(a0); next.push_back(b0); }
            if ((g0 >= 0) != (g1 >= 0)) { const double t = g0 / (g0 - g1); next.push_back(a0 + t * (a1 - a0)); next.push_back(b0 + t * (b1 - b0)); }
        }
        poly.swap(next);
        if (poly.empty()) return false;
    }
    const size_t m = poly.size() / 2;
    double ca = 0, cb = 0;
    for (size_t j = 0; j < m; j++) {
        if (!(fabs(poly[2 * j]) < L * (1.0 - 1e-9) && fabs(poly[2 * j + 1]) < L * (1.0 - 1e-9))) return false;   // reaches the square (or NaN)
        ca += poly[2 * j] / m; cb += poly[2 * j + 1] / m;
    }
    double R = 0;
    for (size_t j = 0; j < m; j++) R = fmax(R, hypot(poly[2 * j] - ca, poly[2 * j + 1] - cb));
    for (int a = 0; a < 3; a++) centre[a] = org[a] + ca * u[a] + cb * w[a];
    // margin: the returned point is o + t d rounded (off the plane by rounding only), to_p is a rounded dot product
    double big = 0;
    for (int k = 0; k < 3 * nv; k++) big = fmax(big, fabs(v[k]));
    *radius = R * (1.0 + 1e-6) + 1e-6 * ext + 1e-9 * big;
    return std::isfinite(*radius);
}

// Axis code of a plane or face for narrow_always' collapsed test, 0 when the solid does not qualify.  Qualifies: the unit
// normal is exactly +-(unit vector k) -> code +-(k + 1); and, for a face, every vertex whose to_next is not zero has an edge
// e_i = v_i - v_i+1 with exactly one non-zero component L, |L| >= 1 and |to_next| >= 1, all coordinates below 1e300.  lo/hi
// receive the coordinate box the vertex tests amount to: vertex i rejects p iff sign(L to_next) (v_i[a] - p[a]) < 0.
int box_code(int kind, const double* n, const FaceV* fv, int nv, double* lo, double* hi) {
    for (int a = 0; a < 3; a++) { lo[a] = -INFINITY; hi[a] = INFINITY; }
    int k = -1;
    for (int a = 0; a < 3; a++) {
        if (n[a] == 1.0 || n[a] == -1.0) { if (k >= 0) return 0; k = a; }
        else if (n[a] != 0.0) return 0;             // also rejects NaN
    }
    if (k < 0) return 0;
    if (kind == RT_KIND_FACE) {
        for (int i = 0; i < nv; i++) {
            const FaceV& f = fv[i];
            for (int a = 0; a < 3; a++) if (!(fabs(f.v[a]) < 1e300) || !(fabs(f.e[a]) < 1e300)) return 0;
            if (!std::isfinite(f.to_next)) return 0;
            if (f.to_next == 0.0) continue;          // (anything) * 0 is never negative: no constraint
            int ax = -1;
            for (int a = 0; a < 3; a++) if (f.e[a] != 0.0) { if (ax >= 0) return 0; ax = a; }
            if (ax < 0) return 0;
            const double L = f.e[ax];
            if (!(fabs(L) >= 1.0) || !(fabs(f.to_next) >= 1.0)) return 0;
            const bool pos = (L > 0) == (f.to_next > 0);
            if (pos) { if (f.v[ax] < hi[ax]) hi[ax] = f.v[ax]; }      // rejects p[ax] > v: p[ax] <= v
            else     { if (f.v[ax] > lo[ax]) lo[ax] = f.v[ax]; }      // rejects p[ax] < v: p[ax] >= v
        }
    }
    return n[k] > 0 ? k + 1 : -(k + 1);
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

// Multiplier and shift for n / d, 0 <= n <= n_max < 2^31: with l = ceil(log2 d) and M = ceil(2^(31 + l) / d) the product n M
// is off from n 2^(31 + l) / d by less than one unit of the quotient, so (n M) >> (31 + l) is the quotient (M < 2^32:
// mul-high, then a shift by l - 1).  Checked here at every multiple of d up to n_max; mul = 0 asks for the division.
inline void div_magic(unsigned d, unsigned long long n_max, unsigned& mul, unsigned& sh) {
    mul = 0; sh = 0;
    if (d < 2 || n_max >= (1ull << 31)) return;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    const unsigned long long M = ((1ull << (31 + l)) + d - 1) / d;
    if (M >> 32) return;
    for (unsigned long long k = 0; k * d <= n_max; k++) {
        const unsigned long long lo = k * d, hi = lo + d - 1 < n_max ? lo + d - 1 : n_max;
        if (((lo * M) >> (31 + l)) != k || ((hi * M) >> (31 + l)) != k) return;
    }
    mul = (unsigned)M; sh = l - 1;
}
// after width, height and tile_rows of kp are final
inline void set_pixel_map(KParams& kp) {
    const unsigned long long pixels = (unsigned long long)(kp.width > 0 ? kp.width : 1) * (unsigned long long)(kp.height > 0 ? kp.height : 1);
    div_magic((unsigned)kp.width, pixels, kp.div_w_mul, kp.div_w_sh);
    div_magic((unsigned)kp.tile_rows, (unsigned long long)(kp.height > 0 ? kp.height : 1), kp.div_t_mul, kp.div_t_sh);
}

void fill_kparams(rt_context* h, KParams& kp) {
    memset(&kp, 0, sizeof(kp));
    kp.width = h->width; kp.height = h->height; kp.spp = h->spp; kp.depth = h->depth;
    kp.scale = h->scale; kp.lens = h->lens; kp.focal = h->focal; kp.frame_seed = h->frame_seed;
    for (int k = 0; k < 3; k++) kp.cam[k] = h->cam[k];
    kp.has_light = h->has_light; kp.light_draws = h->light_draws;
    kp.tile_rows = 1; kp.rank = 0; kp.world = 1;
    set_pixel_map(kp);
    kp.nchunks = h->nchunks;
    kp.n_prims = h->n_prims;
    kp.max_verts = RT_MAXV;
    kp.cmax = h->cmax;
    kp.wide = 0;
    kp.mf_perm = nullptr; kp.mf_cand = nullptr; kp.mf_places_lds = 0;
    for (int k = 0; k < 3; k++) kp.T[k] = h->T[k];
    kp.records = (const float4*)h->d_records;
    kp.chunks = (const ChunkDesc*)h->d_chunks;
    kp.prims = (const PrimD*)h->d_prims;
    kp.verts = (const FaceV*)h->d_verts;
    kp.mats = (const MatD*)h->d_mats;
    kp.always = (const double*)h->d_always; kp.n_always = h->n_always; kp.always_doubles = h->always_doubles; kp.n_box = h->n_box;
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

// fragment table of broad_mfma_kernel for a padded record stream (n_rec a multiple of 32), lengths scaled by s
std::vector<uint4> mf_build_frags(const std::vector<float>& rec, float s) {
    const size_t groups = rec.size() / 4 / BROAD_GROUP;
    std::vector<uint4> fr(groups * MF_GROUP_WORDS);
    for (size_t g = 0; g < groups; g++)
        for (int lane = 0; lane < 64; lane++) fr[g * MF_GROUP_WORDS + lane] = mf_record_frag(&rec[g * 4 * BROAD_GROUP], lane, s);
    return fr;
}

// The scene's table for broad_mfma_kernel.  The records are put in an order of their own -- bounded ones by median cuts
// (mf_kd_order), so that 32 consecutive records are neighbours in space, then the unbounded ones; records
// that are never kept (solids of the always-tested table) are left out -- and `perm` leads from a place in that order
// back to the solid's file index (list entries name places; the exact phase looks the file index up).  Every group of 32
// gets a bounding sphere of its members' bounding spheres; these "group records", 32 to a fragment like ordinary
// records, let the kernel ask first which groups any ray of a tile can touch.  Layout: per LDS window its groups'
// fragments, then the fragments of its group records.
struct MfTable {
    std::vector<uint4> frags;
    std::vector<unsigned> perm;
    int n_groups = 0, n_windows = 0, win_groups = 0;
    float scale = 1.0f;
    double cmax = 0.0;           // largest extent |c| + R over records and group records
    bool reordered = false;      // false: file order, perm is the identity (places are file indices)
    std::vector<float> group_records;   // (centre, R^2 - |c|^2) of every group's bounding sphere; -inf: empty group, +inf: holds an unbounded record
};
inline unsigned mf_spread10(unsigned v) {          // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
#ifndef MF_ORDER_MORTON
#define MF_ORDER_MORTON 0       // 1: the records along a Morton curve through their centres (each axis cut into 1024 steps of its own extent)
#endif
// The order of the bounded records: the set is cut in two across the axis along which its centres spread furthest, at a
// multiple of 32 records nearest the middle, and so on down to groups of 32 -- whatever the solids' arrangement, a group
// is a compact block, consecutive groups are siblings (an LDS window of 32 groups is a block too), and only the last
// group can be short.  (A Morton curve does as well for a 16 x 8 x 8 lattice and worse for a 16 x 12 x 8 one, where its
// power-of-two cuts fall between the lattice's planes unevenly.)  Ties by file index; a group's members in file order.
inline void mf_kd_order(std::vector<int>& idx, int lo, int hi, const float* rec) {
    const int n = hi - lo;
    if (n <= BROAD_GROUP) { std::sort(idx.begin() + lo, idx.begin() + hi); return; }
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = lo; k < hi; k++)
        for (int a = 0; a < 3; a++) { const float v = rec[4 * (size_t)idx[k] + a]; if (v < blo[a]) blo[a] = v; if (v > bhi[a]) bhi[a] = v; }
    int ax = 0;
    for (int a = 1; a < 3; a++) if (bhi[a] - blo[a] > bhi[ax] - blo[ax]) ax = a;
    const int groups = (n + BROAD_GROUP - 1) / BROAD_GROUP;
    const int mid = lo + (groups / 2) * BROAD_GROUP;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int a, int b) {
        const float va = rec[4 * (size_t)a + ax], vb = rec[4 * (size_t)b + ax];
        return va < vb || (va == vb && a < b);
    });
    mf_kd_order(idx, lo, mid, rec);
    mf_kd_order(idx, mid, hi, rec);
}
MfTable mf_build_table(const float* rec_file, int n_prims, double cmax_records, bool reorder) {
    MfTable t;
    t.reordered = reorder;
    std::vector<int> bounded, unbounded;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n_prims; i++) {
        const float* r = rec_file + 4 * (size_t)i;
        if (r[3] == -INFINITY) continue;                       // never kept
        if (!(fabsf(r[3]) <= 3.0e38f)) { unbounded.push_back(i); continue; }
        bounded.push_back(i);
        for (int k = 0; k < 3; k++) { if (r[k] < lo[k]) lo[k] = r[k]; if (r[k] > hi[k]) hi[k] = r[k]; }
    }
    std::vector<std::pair<unsigned, int>> keyed;
#if MF_ORDER_MORTON
    for (int i : bounded) {
        unsigned q[3];
        for (int k = 0; k < 3; k++) {
            const float ext = hi[k] - lo[k];
            const float u = ext > 0.0f ? (rec_file[4 * (size_t)i + k] - lo[k]) / ext : 0.0f;
            q[k] = (unsigned)(u * 1023.0f + 0.5f);
        }
        keyed.push_back({mf_spread10(q[0]) | (mf_spread10(q[1]) << 1) | (mf_spread10(q[2]) << 2), i});
    }
    std::sort(keyed.begin(), keyed.end());
#else
    mf_kd_order(bounded, 0, (int)bounded.size(), rec_file);
    for (size_t k = 0; k < bounded.size(); k++) keyed.push_back({(unsigned)k, bounded[k]});
#endif
    const unsigned NONE = 0x7fffffffu;                        // a place that holds no solid
    if (reorder) {
        for (auto& kv : keyed) t.perm.push_back((unsigned)kv.second);
        for (int i : unbounded) t.perm.push_back((unsigned)i);
    } else {
        for (int i = 0; i < n_prims; i++) t.perm.push_back((unsigned)i);      // never-kept records stay in their places
    }
    while (t.perm.empty() || t.perm.size() % BROAD_GROUP) t.perm.push_back(NONE);
    t.n_groups = (int)(t.perm.size() / BROAD_GROUP);
    // a table in its own order is also scanned by the binned kernel, whose windows hold at most MFB_GROUPS groups
    const int max_win = reorder ? MFB_GROUPS : MF_WINDOW_GROUPS;
    t.n_windows = (t.n_groups + max_win - 1) / max_win;
    t.win_groups = (t.n_groups + t.n_windows - 1) / t.n_windows;
    // records in table order, and the group records
    std::vector<float> rec(t.perm.size() * 4), grp((size_t)t.n_groups * 4);
    t.cmax = cmax_records;
    for (int g = 0; g < t.n_groups; g++) {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        bool any = false, unb = false;
        for (int k = 0; k < BROAD_GROUP; k++) {
            const unsigned i = t.perm[(size_t)g * BROAD_GROUP + k];
            float* r = &rec[((size_t)g * BROAD_GROUP + k) * 4];
            if (i == NONE) { r[0] = r[1] = r[2] = 0.0f; r[3] = -INFINITY; continue; }
            memcpy(r, rec_file + 4 * (size_t)i, 16);
            any = true;
            if (!(fabsf(r[3]) <= 3.0e38f)) { unb = true; continue; }
            const double R = sqrt(fmax(0.0, (double)r[3] + (double)r[0] * r[0] + (double)r[1] * r[1] + (double)r[2] * r[2]));
            for (int a = 0; a < 3; a++) { if (r[a] - R < blo[a]) blo[a] = r[a] - R; if (r[a] + R > bhi[a]) bhi[a] = r[a] + R; }
        }
        float* gr = &grp[(size_t)g * 4];
        if (!any) { gr[0] = gr[1] = gr[2] = 0.0f; gr[3] = -INFINITY; continue; }
        if (unb) { gr[0] = gr[1] = gr[2] = 0.0f; gr[3] = INFINITY; continue; }
        double c[3], R = 0.0;
        for (int a = 0; a < 3; a++) c[a] = 0.5 * (blo[a] + bhi[a]);
        for (int k = 0; k < BROAD_GROUP; k++) {
            const float* r = &rec[((size_t)g * BROAD_GROUP + k) * 4];
            if (r[3] == -INFINITY) continue;
            const double Rm = sqrt(fmax(0.0, (double)r[3] + (double)r[0] * r[0] + (double)r[1] * r[1] + (double)r[2] * r[2]));
            const double d = sqrt((r[0] - c[0]) * (r[0] - c[0]) + (r[1] - c[1]) * (r[1] - c[1]) + (r[2] - c[2]) * (r[2] - c[2])) + Rm;
            if (d > R) R = d;
        }
        const double ext = make_record(c, R * (1.0 + 1e-6) + 1e-30, gr);
        if (ext > t.cmax) t.cmax = ext;
    }
    t.scale = mf_scale_for((float)(t.cmax * (1.0 + 1e-6)));
    t.group_records = grp;
    for (int w = 0; w < t.n_windows; w++) {
        const int g0 = w * t.win_groups, gn = std::min(t.win_groups, t.n_groups - g0);
        for (int g = g0; g < g0 + gn; g++)
            for (int lane = 0; lane < 64; lane++) t.frags.push_back(mf_record_frag(&rec[(size_t)g * 4 * BROAD_GROUP], lane, t.scale));
        for (int l = 0; l * BROAD_GROUP < gn; l++) {
            float g32[4 * BROAD_GROUP];
            for (int k = 0; k < BROAD_GROUP; k++) {
                const int g = l * BROAD_GROUP + k;
                if (g < gn) memcpy(&g32[4 * k], &grp[(size_t)(g0 + g) * 4], 16);
                else { g32[4 * k] = g32[4 * k + 1] = g32[4 * k + 2] = 0.0f; g32[4 * k + 3] = -INFINITY; }
            }
            for (int lane = 0; lane < 64; lane++) t.frags.push_back(mf_record_frag(g32, lane, t.scale));
        }
        // every window but the last is full: the kernel finds window w at w * (win_groups + its group fragments) * 64 words
    }
    return t;
}

template <int BLOCK, bool CULL>
hipError_t mfma_grid(rt_context* h, size_t lds, unsigned* grid) {
    auto kern = broad_mfma_kernel<BLOCK, CULL>;
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
