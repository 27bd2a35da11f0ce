// One process, several GPUs: the frame split into cyclic row tiles over the devices, every device renders, exposes and
// packs its own rows, the packed rows are gathered to device 0 with RCCL (ncclSend / ncclRecv in one group, over xGMI)
// and put into frame order there.  The counterpart in the reference: the threads that share one Color array and render
// cyclic rows (/root/reference/src/main.cpp:250,303-313), then expose() (:314) and writeImage (:327).
// A section of librt_hip.so's single translation unit: rt_device.hip includes it after the C ABI it builds on.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <thread>

namespace {

// RCCL is looked up when the first multi-GPU context is created: a process that renders on one GPU never loads it, and
// a process that already holds a copy (PyTorch brings its own) keeps using that one.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string load() {
        if (lib) return "";
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return std::string("librccl.so not loadable: ") + dlerror();
#define RT_RCCL_SYM(field, sym)                                                \
        field = (decltype(field))dlsym(lib, sym);                              \
        if (!field) { lib = nullptr; return std::string("RCCL lacks ") + sym; }
        RT_RCCL_SYM(CommInitAll, "ncclCommInitAll") RT_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RT_RCCL_SYM(GroupStart, "ncclGroupStart") RT_RCCL_SYM(GroupEnd, "ncclGroupEnd")
        RT_RCCL_SYM(Send, "ncclSend") RT_RCCL_SYM(Recv, "ncclRecv") RT_RCCL_SYM(AllGather, "ncclAllGather")
        RT_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_RCCL_SYM
        return "";
    }
};
RcclApi g_rccl;

// frame row y <- row of the gathered buffer [world][pad_rows][W] it was packed into by its rank
__host__ __device__ inline int gather_source_row(int tile_rows, int world, int pad_rows, int y) {
    const int tile = y / tile_rows;
    return (tile % world) * pad_rows + (tile / world) * tile_rows + y % tile_rows;
}

__global__ void deinterleave_rows_kernel(const int* __restrict__ gathered, int* __restrict__ frame, int W, int H, int tile_rows,
                                         int world, int pad_rows) {
    const size_t n = (size_t)W * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)W), x = (int)(i - (size_t)y * W);
        frame[i] = gathered[(size_t)gather_source_row(tile_rows, world, pad_rows, y) * W + x];
    }
}

}  // namespace

struct rt_multi_context {
    int n = 0;                        // ranks = devices used (loopback: ranks that all live on device 0)
    bool loopback = false;            // every rank on device 0, rows moved by device copies instead of RCCL (one-GPU tests)
    std::vector<int> device;          // rank -> HIP device
    std::vector<rt_handle> ctx;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;     // empty in loopback mode and for a single rank
    // per rank: radiance rows, packed rows, over-exposure bits and every rank's bits; on rank 0 the gathered rows and the frame
    std::vector<double*> d_rgb;
    std::vector<int32_t*> d_img;
    std::vector<uint8_t*> d_bits, d_bits_all;
    int32_t* d_gathered = nullptr;
    int32_t* d_frame = nullptr;
    int buf_w = 0, buf_h = 0, buf_tile = 0;
    std::vector<hipEvent_t> done;     // per rank: its rows are packed (loopback ordering)
    std::string err;
    bool have_scene = false;
    int width = 0, height = 0, num_blurs = 0;
    double exposure = 1;
    double last_ms = 0;
};

namespace {

int mfail(rt_multi_context* m, int code, const std::string& msg) { m->err = msg; return code; }

#define MHIP_TRY(m, call)                                                                                         \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) return mfail(m, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)
#define MNCCL_TRY(m, call)                                                                                        \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) return mfail(m, RT_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

void multi_free_buffers(rt_multi_context* m) {
    for (int r = 0; r < m->n; r++) {
        hipSetDevice(m->device[r]);
        if (r < (int)m->d_rgb.size()) { hipFree(m->d_rgb[r]); hipFree(m->d_img[r]); hipFree(m->d_bits[r]); hipFree(m->d_bits_all[r]); }
    }
    m->d_rgb.clear(); m->d_img.clear(); m->d_bits.clear(); m->d_bits_all.clear();
    if (m->n) hipSetDevice(m->device[0]);
    hipFree(m->d_gathered); hipFree(m->d_frame);
    m->d_gathered = nullptr; m->d_frame = nullptr;
    m->buf_w = m->buf_h = m->buf_tile = 0;
}

int multi_pad_rows(int height, int tile_rows, int world) {
    int pad = 0;
    for (int r = 0; r < world; r++) { const int k = rt_rows_for_rank(height, tile_rows, r, world); if (k > pad) pad = k; }
    return pad > 0 ? pad : 1;
}

int multi_ensure_buffers(rt_multi_context* m, int tile_rows) {
    if (m->buf_w == m->width && m->buf_h == m->height && m->buf_tile == tile_rows) return RT_OK;
    multi_free_buffers(m);
    const size_t W = (size_t)m->width;
    const size_t pad = (size_t)multi_pad_rows(m->height, tile_rows, m->n);
    m->d_rgb.assign(m->n, nullptr); m->d_img.assign(m->n, nullptr); m->d_bits.assign(m->n, nullptr); m->d_bits_all.assign(m->n, nullptr);
    for (int r = 0; r < m->n; r++) {
        MHIP_TRY(m, hipSetDevice(m->device[r]));
        MHIP_TRY(m, hipMalloc((void**)&m->d_rgb[r], pad * W * 3 * sizeof(double)));
        MHIP_TRY(m, hipMalloc((void**)&m->d_img[r], pad * W * sizeof(int32_t)));
        MHIP_TRY(m, hipMalloc((void**)&m->d_bits[r], pad * W));
        MHIP_TRY(m, hipMalloc((void**)&m->d_bits_all[r], (size_t)m->n * pad * W));
    }
    MHIP_TRY(m, hipSetDevice(m->device[0]));
    MHIP_TRY(m, hipMalloc((void**)&m->d_gathered, (size_t)m->n * pad * W * sizeof(int32_t)));
    MHIP_TRY(m, hipMalloc((void**)&m->d_frame, W * (size_t)m->height * sizeof(int32_t)));
    m->buf_w = m->width; m->buf_h = m->height; m->buf_tile = tile_rows;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_gather_source_row(int height, int tile_rows, int world, int pad_rows, int y) {
    if (height < 1 || tile_rows < 1 || world < 1 || pad_rows < 1 || y < 0 || y >= height) return RT_ERR_ARG;
    return gather_source_row(tile_rows, world, pad_rows, y);
}

int rt_multi_create(rt_multi_handle* out, int n_gpus) {
    if (!out) return RT_ERR_ARG;
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return RT_ERR_NO_DEVICE;
    // RT_MULTI_LOOPBACK=1: n_gpus ranks that all live on device 0 and exchange rows by device copies -- the whole
    // multi-rank path except the RCCL calls, for machines with one GPU
    const char* lb = getenv("RT_MULTI_LOOPBACK");
    const bool loopback = lb && atoi(lb) != 0;
    if (n_gpus < 1 || n_gpus > 64 || (!loopback && n_gpus > have)) return RT_ERR_ARG;
    rt_multi_context* m = new rt_multi_context();
    m->loopback = loopback;
    for (int r = 0; r < n_gpus; r++) {
        const int dev = loopback ? 0 : r;
        rt_handle h = nullptr;
        const int rc = rt_create(&h, dev);
        if (rc != RT_OK) { rt_multi_destroy(m); return rc; }
        m->device.push_back(dev); m->ctx.push_back(h); m->n++;
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { if (st) hipStreamDestroy(st); rt_multi_destroy(m); return RT_ERR_HIP; }
        m->stream.push_back(st); m->done.push_back(ev);
    }
    if (n_gpus > 1 && !loopback) {
        const std::string e = g_rccl.load();
        if (!e.empty()) { rt_multi_destroy(m); return RT_ERR_NO_DEVICE; }
        m->comm.assign(n_gpus, nullptr);
        if (g_rccl.CommInitAll(m->comm.data(), n_gpus, m->device.data()) != ncclSuccess) { m->comm.clear(); rt_multi_destroy(m); return RT_ERR_HIP; }
    }
    *out = m;
    return RT_OK;
}

void rt_multi_destroy(rt_multi_handle m) {
    if (!m) return;
    for (size_t r = 0; r < m->stream.size(); r++) { hipSetDevice(m->device[r]); hipStreamSynchronize(m->stream[r]); }
    for (ncclComm_t c : m->comm) if (c) g_rccl.CommDestroy(c);
    multi_free_buffers(m);
    for (size_t r = 0; r < m->stream.size(); r++) { hipSetDevice(m->device[r]); hipStreamDestroy(m->stream[r]); hipEventDestroy(m->done[r]); }
    for (rt_handle h : m->ctx) rt_destroy(h);
    delete m;
}

const char* rt_multi_last_error(rt_multi_handle m) { return m ? m->err.c_str() : "null handle"; }

int rt_multi_gpus(rt_multi_handle m) { return m ? m->n : RT_ERR_ARG; }

int rt_multi_set_option(rt_multi_handle m, int key, int value) {
    if (!m) return RT_ERR_ARG;
    for (rt_handle h : m->ctx) {
        const int rc = rt_set_option(h, key, value);
        if (rc != RT_OK) return mfail(m, rc, rt_last_error(h));
    }
    return RT_OK;
}

int rt_multi_upload_scene(rt_multi_handle m, const double* params, int n_prims, const double* prims, int n_mats,
                          const double* mats, int n_verts, const double* verts) {
    if (!m) return RT_ERR_ARG;
    m->have_scene = false;
    for (rt_handle h : m->ctx) {
        const int rc = rt_upload_scene(h, params, n_prims, prims, n_mats, mats, n_verts, verts);
        if (rc != RT_OK) return mfail(m, rc, rt_last_error(h));
    }
    m->width = m->ctx[0]->width; m->height = m->ctx[0]->height;
    m->num_blurs = m->ctx[0]->num_blurs; m->exposure = m->ctx[0]->exposure;
    m->have_scene = true;
    return RT_OK;
}

int rt_multi_render_image(rt_multi_handle m, int tile_rows, int32_t* img, uint64_t* rays) {
    if (!m) return RT_ERR_ARG;
    if (!m->have_scene) return mfail(m, RT_ERR_NO_SCENE, "rt_multi_render_image: no scene uploaded");
    if (!img || tile_rows < 0) return mfail(m, RT_ERR_ARG, "rt_multi_render_image: bad argument");
    const int N = m->n, W = m->width, H = m->height;
    // default partition: 8-row tiles dealt out cyclically; with blur passes one band per rank (a rank then blooms only
    // the block rows around its band instead of nearly the whole frame)
    if (tile_rows == 0) tile_rows = (m->num_blurs > 0 && N > 1) ? (H + N - 1) / N : 8;
    { const int rc = multi_ensure_buffers(m, tile_rows); if (rc != RT_OK) return rc; }
    const int pad = multi_pad_rows(H, tile_rows, N);
    const auto t_start = std::chrono::steady_clock::now();

    // every rank: render its rows, then pack them (no blur pass) or reduce them to over-exposure bits, on its own stream.
    // One host thread per rank enqueues (a frame is some forty launches per rank).
    std::vector<int> rc(N, RT_OK);
    auto render_rank = [&](int r) {
        rt_handle h = m->ctx[r];
        const int rows = rt_rows_for_rank(H, tile_rows, r, N);
        rc[r] = rt_render_rows_device(h, tile_rows, r, N, m->d_rgb[r], nullptr, m->stream[r]);
        if (rc[r] != RT_OK || rows == 0) return;
        if (m->num_blurs == 0 || N == 1) rc[r] = rt_expose_device(h, m->d_rgb[r], W, rows, m->exposure, N == 1 ? m->num_blurs : 0, m->d_img[r], m->stream[r]);
        else rc[r] = rt_expose_bits_device(h, m->d_rgb[r], W, rows, m->exposure, m->d_bits[r], m->stream[r]);
    };
    {
        std::vector<std::thread> th;
        for (int r = 1; r < N; r++) th.emplace_back(render_rank, r);
        render_rank(0);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < N; r++) if (rc[r] != RT_OK) return mfail(m, rc[r], rt_last_error(m->ctx[r]));

    if (m->num_blurs > 0 && N > 1) {
        // all-gather of the bits (1 byte per pixel), then every rank blooms what its rows depend on and packs its rows
        const size_t cnt = (size_t)pad * W;
        if (m->loopback) {
            for (int r = 0; r < N; r++) MHIP_TRY(m, hipEventRecord(m->done[r], m->stream[r]));
            for (int r = 0; r < N; r++) {
                for (int s = 0; s < N; s++) {
                    if (s != r) MHIP_TRY(m, hipStreamWaitEvent(m->stream[r], m->done[s], 0));
                    MHIP_TRY(m, hipMemcpyAsync(m->d_bits_all[r] + (size_t)s * cnt, m->d_bits[s], cnt, hipMemcpyDeviceToDevice, m->stream[r]));
                }
            }
        } else {
            MNCCL_TRY(m, g_rccl.GroupStart());
            for (int r = 0; r < N; r++) MNCCL_TRY(m, g_rccl.AllGather(m->d_bits[r], m->d_bits_all[r], cnt, ncclUint8, m->comm[r], m->stream[r]));
            MNCCL_TRY(m, g_rccl.GroupEnd());
        }
        for (int r = 0; r < N; r++) {
            const int rcr = rt_expose_rows_device(m->ctx[r], m->d_rgb[r], m->d_bits_all[r], W, H, tile_rows, r, N, pad, m->exposure, m->num_blurs,
                                                  m->d_img[r], m->stream[r]);
            if (rcr != RT_OK) return mfail(m, rcr, rt_last_error(m->ctx[r]));
        }
    }

    // gather of the packed rows to rank 0: 4 bytes per pixel in all, each rank's rows over its own link into device 0
    const size_t cnt = (size_t)pad * W;
    MHIP_TRY(m, hipSetDevice(m->device[0]));
    MHIP_TRY(m, hipMemcpyAsync(m->d_gathered, m->d_img[0], cnt * sizeof(int32_t), hipMemcpyDeviceToDevice, m->stream[0]));
    if (N > 1) {
        if (m->loopback) {
            for (int r = 1; r < N; r++) {
                MHIP_TRY(m, hipEventRecord(m->done[r], m->stream[r]));
                MHIP_TRY(m, hipStreamWaitEvent(m->stream[0], m->done[r], 0));
                MHIP_TRY(m, hipMemcpyAsync(m->d_gathered + (size_t)r * cnt, m->d_img[r], cnt * sizeof(int32_t), hipMemcpyDeviceToDevice, m->stream[0]));
            }
        } else {
            MNCCL_TRY(m, g_rccl.GroupStart());
            for (int r = 1; r < N; r++) {
                MNCCL_TRY(m, g_rccl.Send(m->d_img[r], cnt, ncclInt32, 0, m->comm[r], m->stream[r]));
                MNCCL_TRY(m, g_rccl.Recv(m->d_gathered + (size_t)r * cnt, cnt, ncclInt32, r, m->comm[0], m->stream[0]));
            }
            MNCCL_TRY(m, g_rccl.GroupEnd());
        }
    }
    const size_t npx = (size_t)W * H;
    unsigned grid = (unsigned)((npx + 255) / 256);
    if (grid > 8192u) grid = 8192u;
    hipLaunchKernelGGL(deinterleave_rows_kernel, dim3(grid), dim3(256), 0, m->stream[0], (const int*)m->d_gathered, (int*)m->d_frame, W, H,
                       tile_rows, N, pad);
    MHIP_TRY(m, hipGetLastError());
    MHIP_TRY(m, hipMemcpyAsync(img, m->d_frame, npx * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream[0]));
    MHIP_TRY(m, hipStreamSynchronize(m->stream[0]));
    uint64_t total = 0;
    for (int r = 0; r < N; r++) {
        uint64_t k = 0;
        MHIP_TRY(m, hipSetDevice(m->device[r]));
        MHIP_TRY(m, hipStreamSynchronize(m->stream[r]));
        const int rcr = rt_finish(m->ctx[r], &k);
        if (rcr != RT_OK) return mfail(m, rcr, rt_last_error(m->ctx[r]));
        total += k;
    }
    if (rays) *rays = total;
    m->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    return RT_OK;
}

double rt_multi_last_ms(rt_multi_handle m) { return m ? m->last_ms : 0.0; }

}  // extern "C"
