// One process, several GPUs: the frame split into cyclic row tiles over the devices, every device renders, exposes and
// packs its own rows, the packed rows are gathered to device 0 with RCCL (ncclSend / ncclRecv in one group, over xGMI)
// and put into frame order there.  The counterpart in the reference: the threads that share one Color array and render
// cyclic rows (/root/reference/src/main.cpp:250,303-313), then expose() (:314) and writeImage (:327).
//
// Host side: one persistent worker thread per rank beyond the first enqueues that rank's launches (a frame is some
// forty launches per rank; one thread for eight ranks would take longer to enqueue a frame than the GPUs take to
// render it), the calling thread enqueues rank 0's.  A frame is begun (everything enqueued, nothing waited for) and
// ended (waited for, copied out) by separate calls and two frames may be in flight: the gather, the re-ordering and
// the copy to the host of frame i run on the ranks' exchange streams while their render streams are already busy with
// frame i + 1.  Per rank two buffers of packed rows, on rank 0 two gathered buffers, two frames and two pinned host frames.
// A section of librt_hip.so's single translation unit: rt_device.hip includes it after the C ABI it builds on.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <thread>

namespace {

// RCCL is looked up when the first context that exchanges through it is created: a process that renders on one GPU
// never loads it, and a process that already holds a copy (PyTorch brings its own) keeps using that one.
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
        std::string why;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
            const char* e = dlerror();
            why += std::string(why.empty() ? "" : "; ") + (e ? e : name);
        }
        if (!lib) return "librccl.so not loadable: " + why;
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
std::string g_multi_create_err;        // why the last rt_multi_create failed (there is no handle to keep it in)

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

constexpr int MULTI_SLOTS = 2;         // frames in flight

}  // namespace

struct rt_multi_context {
    int n = 0;                        // ranks = devices used (loopback: ranks that all live on device 0)
    bool loopback = false;            // every rank on device 0, rows moved by device copies instead of RCCL (one-GPU tests)
    bool use_rccl = false;            // rows and bits move through RCCL (n > 1 on n devices; or forced, for one rank)
    std::vector<int> device;          // rank -> HIP device
    std::vector<rt_handle> ctx;
    std::vector<hipStream_t> stream;  // per rank: render, expose, pack
    std::vector<hipStream_t> xstream; // per rank: the exchange with rank 0 (and, on rank 0, frame order + copy to the host)
    std::vector<ncclComm_t> comm;     // empty unless use_rccl
    // per rank: radiance rows, over-exposure bits and every rank's bits; per rank and frame slot: packed rows
    std::vector<double*> d_rgb;
    std::vector<uint8_t*> d_bits, d_bits_all;
    std::vector<int32_t*> d_img[MULTI_SLOTS];
    // rank 0, per frame slot: the gathered rows, the frame, the frame on the host (pinned)
    int32_t* d_gathered[MULTI_SLOTS] = {};
    int32_t* d_frame[MULTI_SLOTS] = {};
    int32_t* h_frame[MULTI_SLOTS] = {};
    unsigned* h_ctrl[MULTI_SLOTS] = {};      // pinned, [n][8]: every rank's control words (error bits, ray counter) of the frame
    int buf_w = 0, buf_h = 0, buf_tile = 0;
    // per frame slot and rank: rows packed (render stream); bits ready (render stream; loopback all-gather); rows sent
    // (exchange stream); per slot: frame on the host
    std::vector<hipEvent_t> packed[MULTI_SLOTS], bits_ready[MULTI_SLOTS], sent[MULTI_SLOTS];
    hipEvent_t frame_done[MULTI_SLOTS] = {};
    bool in_flight[MULTI_SLOTS] = {};
    int slot_tile[MULTI_SLOTS] = {};
    std::chrono::steady_clock::time_point slot_t0[MULTI_SLOTS];
    double slot_enqueue_ms[MULTI_SLOTS] = {};
    unsigned next_slot = 0, oldest_slot = 0; // frames begin and end in order
    int frames_in_flight = 0;
    std::string err;
    bool have_scene = false;
    int width = 0, height = 0, num_blurs = 0;
    double exposure = 1;
    double last_ms = 0, last_enqueue_ms = 0, last_wait_ms = 0;
    // the workers (ranks 1 .. n-1): a job is "enqueue your part of a frame"
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t job_gen = 0;
    int job_pending = 0, job_tile_rows = 0, job_slot = 0;
    bool quit = false;
    std::vector<int> job_rc;
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
        if (r < (int)m->d_rgb.size()) {
            hipFree(m->d_rgb[r]); hipFree(m->d_bits[r]); hipFree(m->d_bits_all[r]);
            for (int s = 0; s < MULTI_SLOTS; s++) hipFree(m->d_img[s][r]);
        }
    }
    m->d_rgb.clear(); m->d_bits.clear(); m->d_bits_all.clear();
    if (m->n) hipSetDevice(m->device[0]);
    for (int s = 0; s < MULTI_SLOTS; s++) {
        m->d_img[s].clear();
        hipFree(m->d_gathered[s]); hipFree(m->d_frame[s]);
        if (m->h_frame[s]) hipHostFree(m->h_frame[s]);
        m->d_gathered[s] = m->d_frame[s] = m->h_frame[s] = nullptr;
    }
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
    m->d_rgb.assign(m->n, nullptr); m->d_bits.assign(m->n, nullptr); m->d_bits_all.assign(m->n, nullptr);
    for (int s = 0; s < MULTI_SLOTS; s++) m->d_img[s].assign(m->n, nullptr);
    for (int r = 0; r < m->n; r++) {
        MHIP_TRY(m, hipSetDevice(m->device[r]));
        MHIP_TRY(m, hipMalloc((void**)&m->d_rgb[r], pad * W * 3 * sizeof(double)));
        for (int s = 0; s < MULTI_SLOTS; s++) MHIP_TRY(m, hipMalloc((void**)&m->d_img[s][r], pad * W * sizeof(int32_t)));
        MHIP_TRY(m, hipMalloc((void**)&m->d_bits[r], pad * W));
        MHIP_TRY(m, hipMalloc((void**)&m->d_bits_all[r], (size_t)m->n * pad * W));
    }
    MHIP_TRY(m, hipSetDevice(m->device[0]));
    for (int s = 0; s < MULTI_SLOTS; s++) {
        MHIP_TRY(m, hipMalloc((void**)&m->d_gathered[s], (size_t)m->n * pad * W * sizeof(int32_t)));
        MHIP_TRY(m, hipMalloc((void**)&m->d_frame[s], W * (size_t)m->height * sizeof(int32_t)));
        MHIP_TRY(m, hipHostMalloc((void**)&m->h_frame[s], W * (size_t)m->height * sizeof(int32_t), hipHostMallocDefault));
    }
    m->buf_w = m->width; m->buf_h = m->height; m->buf_tile = tile_rows;
    return RT_OK;
}

// Rank r's own part of a frame, enqueued on its render stream: render its rows, copy out its control words, then pack
// the rows (no blur pass, or a single rank: the whole expose) or reduce them to over-exposure bits.
int multi_enqueue_rank(rt_multi_context* m, int r, int tile_rows, int slot) {
    rt_handle h = m->ctx[r];
    const int N = m->n, W = m->width, H = m->height;
    const int rows = rt_rows_for_rank(H, tile_rows, r, N);
    if (hipSetDevice(m->device[r]) != hipSuccess) return fail(h, RT_ERR_HIP, "hipSetDevice failed");
    int rc = rt_render_rows_device(h, tile_rows, r, N, m->d_rgb[r], nullptr, m->stream[r]);
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(m->h_ctrl[slot] + 8 * r, h->d_ctrl, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, m->stream[r]));
    if (rows > 0) {
        if (m->num_blurs == 0 || N == 1) rc = rt_expose_device(h, m->d_rgb[r], W, rows, m->exposure, N == 1 ? m->num_blurs : 0, m->d_img[slot][r], m->stream[r]);
        else rc = rt_expose_bits_device(h, m->d_rgb[r], W, rows, m->exposure, m->d_bits[r], m->stream[r]);
        if (rc != RT_OK) return rc;
    }
    if (m->num_blurs > 0 && N > 1) HIP_TRY(h, hipEventRecord(m->bits_ready[slot][r], m->stream[r]));
    else HIP_TRY(h, hipEventRecord(m->packed[slot][r], m->stream[r]));
    return RT_OK;
}

void multi_worker(rt_multi_context* m, int r) {
    hipSetDevice(m->device[r]);
    uint64_t seen = 0;
    for (;;) {
        int tile_rows, slot;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_job.wait(lk, [&] { return m->quit || m->job_gen != seen; });
            if (m->quit) return;
            seen = m->job_gen; tile_rows = m->job_tile_rows; slot = m->job_slot;
        }
        const int rc = multi_enqueue_rank(m, r, tile_rows, slot);
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->job_rc[r] = rc;
            if (--m->job_pending == 0) m->cv_done.notify_all();
        }
    }
}

// after a failed enqueue: let every stream drain so that no launch of the failed frame is still reading a buffer the
// caller may free or reuse, and forget the frames in flight
void multi_drain(rt_multi_context* m) {
    for (int r = 0; r < m->n; r++) {
        hipSetDevice(m->device[r]);
        hipStreamSynchronize(m->stream[r]);
        hipStreamSynchronize(m->xstream[r]);
        m->ctx[r]->events_pending = false;
    }
    for (int s = 0; s < MULTI_SLOTS; s++) m->in_flight[s] = false;
    m->frames_in_flight = 0;
    m->oldest_slot = m->next_slot;
}

int multi_begin(rt_multi_context* m, int tile_rows) {
    const int N = m->n, W = m->width, H = m->height;
    // default partition: 8-row tiles dealt out cyclically; with blur passes one band per rank (a rank then blooms only
    // the block rows around its band instead of nearly the whole frame)
    if (tile_rows == 0) tile_rows = (m->num_blurs > 0 && N > 1) ? (H + N - 1) / N : 8;
    if (m->frames_in_flight > 0 && (m->buf_w != W || m->buf_h != H || m->buf_tile != tile_rows))
        return mfail(m, RT_ERR_ARG, "rt_multi_render_begin: a frame with another partition is still in flight");
    { const int rc = multi_ensure_buffers(m, tile_rows); if (rc != RT_OK) return rc; }
    const int pad = multi_pad_rows(H, tile_rows, N);
    const int slot = (int)(m->next_slot % MULTI_SLOTS);
    const auto t0 = std::chrono::steady_clock::now();

    // every rank enqueues its own part: the workers theirs, this thread rank 0's
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->job_tile_rows = tile_rows; m->job_slot = slot; m->job_pending = N - 1;
        for (int r = 0; r < N; r++) m->job_rc[r] = RT_OK;
        m->job_gen++;
    }
    if (N > 1) m->cv_job.notify_all();
    m->job_rc[0] = multi_enqueue_rank(m, 0, tile_rows, slot);
    if (N > 1) {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { return m->job_pending == 0; });
    }
    for (int r = 0; r < N; r++)
        if (m->job_rc[r] != RT_OK) { const int rc = m->job_rc[r]; const std::string e = rt_last_error(m->ctx[r]); multi_drain(m); return mfail(m, rc, e); }

    // from here on an error leaves launches behind on several streams: drain them before reporting it
    const int rc_x = [&]() -> int {
        const size_t cnt = (size_t)pad * W;
        if (m->num_blurs > 0 && N > 1) {
            // all-gather of the bits (1 byte per pixel) on the render streams, then every rank blooms what its rows depend on
            // and packs its rows
            if (m->use_rccl) {
                MNCCL_TRY(m, g_rccl.GroupStart());
                for (int r = 0; r < N; r++) MNCCL_TRY(m, g_rccl.AllGather(m->d_bits[r], m->d_bits_all[r], cnt, ncclUint8, m->comm[r], m->stream[r]));
                MNCCL_TRY(m, g_rccl.GroupEnd());
            } else {
                for (int r = 0; r < N; r++) {
                    MHIP_TRY(m, hipSetDevice(m->device[r]));
                    for (int s = 0; s < N; s++) {
                        if (s != r) MHIP_TRY(m, hipStreamWaitEvent(m->stream[r], m->bits_ready[slot][s], 0));
                        MHIP_TRY(m, hipMemcpyAsync(m->d_bits_all[r] + (size_t)s * cnt, m->d_bits[s], cnt, hipMemcpyDeviceToDevice, m->stream[r]));
                    }
                }
            }
            for (int r = 0; r < N; r++) {
                const int rcr = rt_expose_rows_device(m->ctx[r], m->d_rgb[r], m->d_bits_all[r], W, H, tile_rows, r, N, pad, m->exposure, m->num_blurs,
                                                      m->d_img[slot][r], m->stream[r]);
                if (rcr != RT_OK) return mfail(m, rcr, rt_last_error(m->ctx[r]));
                MHIP_TRY(m, hipSetDevice(m->device[r]));
                MHIP_TRY(m, hipEventRecord(m->packed[slot][r], m->stream[r]));
            }
        } else if (m->num_blurs > 0 && m->use_rccl) {
            // one rank whose exchange is forced through RCCL: its bits make the round trip as well (the frame is exposed whole
            // by rt_expose_device; the gathered bits are not needed for it)
            MHIP_TRY(m, hipSetDevice(m->device[0]));
            const int rcb = rt_expose_bits_device(m->ctx[0], m->d_rgb[0], W, rt_rows_for_rank(H, tile_rows, 0, 1), m->exposure, m->d_bits[0], m->stream[0]);
            if (rcb != RT_OK) return mfail(m, rcb, rt_last_error(m->ctx[0]));
            MNCCL_TRY(m, g_rccl.GroupStart());
            MNCCL_TRY(m, g_rccl.AllGather(m->d_bits[0], m->d_bits_all[0], cnt, ncclUint8, m->comm[0], m->stream[0]));
            MNCCL_TRY(m, g_rccl.GroupEnd());
        }

        // gather of the packed rows to rank 0 on the exchange streams: 4 bytes per pixel in all, each rank's rows over its
        // own link into device 0; the render streams are free for the next frame meanwhile
        for (int r = 0; r < N; r++) {
            MHIP_TRY(m, hipSetDevice(m->device[r]));
            MHIP_TRY(m, hipStreamWaitEvent(m->xstream[r], m->packed[slot][r], 0));
            if (!m->use_rccl && r > 0) MHIP_TRY(m, hipStreamWaitEvent(m->xstream[0], m->packed[slot][r], 0));
        }
        MHIP_TRY(m, hipSetDevice(m->device[0]));
        if (m->use_rccl) {
            MNCCL_TRY(m, g_rccl.GroupStart());
            for (int r = 0; r < N; r++) {
                MNCCL_TRY(m, g_rccl.Send(m->d_img[slot][r], cnt, ncclInt32, 0, m->comm[r], m->xstream[r]));
                MNCCL_TRY(m, g_rccl.Recv(m->d_gathered[slot] + (size_t)r * cnt, cnt, ncclInt32, r, m->comm[0], m->xstream[0]));
            }
            MNCCL_TRY(m, g_rccl.GroupEnd());
        } else {
            for (int r = 0; r < N; r++)
                MHIP_TRY(m, hipMemcpyAsync(m->d_gathered[slot] + (size_t)r * cnt, m->d_img[slot][r], cnt * sizeof(int32_t), hipMemcpyDeviceToDevice, m->xstream[0]));
        }
        for (int r = 1; r < N; r++) {
            MHIP_TRY(m, hipSetDevice(m->device[r]));
            MHIP_TRY(m, hipEventRecord(m->sent[slot][r], m->xstream[r]));
        }
        MHIP_TRY(m, hipSetDevice(m->device[0]));
        const size_t npx = (size_t)W * H;
        unsigned grid = (unsigned)((npx + 255) / 256);
        if (grid > 8192u) grid = 8192u;
        hipLaunchKernelGGL(deinterleave_rows_kernel, dim3(grid), dim3(256), 0, m->xstream[0], (const int*)m->d_gathered[slot], (int*)m->d_frame[slot], W, H,
                           tile_rows, N, pad);
        MHIP_TRY(m, hipGetLastError());
        MHIP_TRY(m, hipMemcpyAsync(m->h_frame[slot], m->d_frame[slot], npx * sizeof(int32_t), hipMemcpyDeviceToHost, m->xstream[0]));
        MHIP_TRY(m, hipEventRecord(m->frame_done[slot], m->xstream[0]));
        return RT_OK;
    }();
    if (rc_x != RT_OK) { multi_drain(m); return rc_x; }
    m->in_flight[slot] = true; m->slot_tile[slot] = tile_rows; m->slot_t0[slot] = t0;
    m->slot_enqueue_ms[slot] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->next_slot++; m->frames_in_flight++;
    return RT_OK;
}

int multi_end(rt_multi_context* m, int32_t* img, uint64_t* rays) {
    const int N = m->n;
    const int slot = (int)(m->oldest_slot % MULTI_SLOTS);
    const auto tw = std::chrono::steady_clock::now();
    const int rc_w = [&]() -> int {
        MHIP_TRY(m, hipSetDevice(m->device[0]));
        MHIP_TRY(m, hipEventSynchronize(m->frame_done[slot]));
        for (int r = 1; r < N; r++) {      // the senders have let go of their packed rows (the slot is free again)
            MHIP_TRY(m, hipSetDevice(m->device[r]));
            MHIP_TRY(m, hipEventSynchronize(m->sent[slot][r]));
        }
        return RT_OK;
    }();
    if (rc_w != RT_OK) { multi_drain(m); return rc_w; }
    m->in_flight[slot] = false; m->oldest_slot++; m->frames_in_flight--;
    if (m->frames_in_flight == 0) for (int r = 0; r < N; r++) m->ctx[r]->events_pending = false;
    uint64_t total = 0;
    for (int r = 0; r < N; r++) {
        const unsigned* c = m->h_ctrl[slot] + 8 * r;
        total += (uint64_t)c[2] | ((uint64_t)c[3] << 32);
        const int rcg = guard_status(m->ctx[r], c);
        if (rcg != RT_OK) return mfail(m, rcg, rt_last_error(m->ctx[r]));
    }
    memcpy(img, m->h_frame[slot], (size_t)m->width * m->height * sizeof(int32_t));
    if (rays) *rays = total;
    const auto now = std::chrono::steady_clock::now();
    m->last_ms = std::chrono::duration<double, std::milli>(now - m->slot_t0[slot]).count();
    m->last_enqueue_ms = m->slot_enqueue_ms[slot];
    m->last_wait_ms = std::chrono::duration<double, std::milli>(now - tw).count();
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
    g_multi_create_err.clear();
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { g_multi_create_err = "no HIP device"; return RT_ERR_NO_DEVICE; }
    // RT_MULTI_LOOPBACK=1: n_gpus ranks that all live on device 0 and exchange rows by device copies -- the whole
    // multi-rank path except the RCCL calls, for machines with one GPU.  RT_MULTI_FORCE_RCCL=1: a context of ONE rank
    // creates a communicator all the same and sends its rows (and bits) to itself through RCCL -- the library's RCCL
    // calls on a machine with one GPU.
    const char* lb = getenv("RT_MULTI_LOOPBACK");
    const bool loopback = lb && atoi(lb) != 0;
    const char* fr = getenv("RT_MULTI_FORCE_RCCL");
    const bool force_rccl = fr && atoi(fr) != 0 && !loopback;
    if (n_gpus < 1 || n_gpus > 64 || (!loopback && n_gpus > have)) { g_multi_create_err = "rt_multi_create: n_gpus must be 1 .. the number of HIP devices"; return RT_ERR_ARG; }
    rt_multi_context* m = new rt_multi_context();
    m->loopback = loopback;
    auto give_up = [&](int code, const std::string& why) { g_multi_create_err = why; rt_multi_destroy(m); return code; };
    for (int r = 0; r < n_gpus; r++) {
        const int dev = loopback ? 0 : r;
        rt_handle h = nullptr;
        const int rc = rt_create(&h, dev);
        if (rc != RT_OK) return give_up(rc, "rt_create failed on device " + std::to_string(dev));
        m->device.push_back(dev); m->ctx.push_back(h); m->n++;
        hipStream_t st = nullptr, xs = nullptr;
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) != hipSuccess) {
            if (st) hipStreamDestroy(st);
            return give_up(RT_ERR_HIP, "stream creation failed on device " + std::to_string(dev));
        }
        m->stream.push_back(st); m->xstream.push_back(xs);
        for (int s = 0; s < MULTI_SLOTS; s++) {
            hipEvent_t a = nullptr, b = nullptr, c = nullptr;
            if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c, hipEventDisableTiming) != hipSuccess) {
                if (a) hipEventDestroy(a);
                if (b) hipEventDestroy(b);
                return give_up(RT_ERR_HIP, "event creation failed");
            }
            m->packed[s].push_back(a); m->bits_ready[s].push_back(b); m->sent[s].push_back(c);
        }
    }
    hipSetDevice(m->device[0]);
    for (int s = 0; s < MULTI_SLOTS; s++) {
        if (hipEventCreateWithFlags(&m->frame_done[s], hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void**)&m->h_ctrl[s], (size_t)n_gpus * 8 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess)
            return give_up(RT_ERR_HIP, "event or pinned-memory creation failed");
        memset(m->h_ctrl[s], 0, (size_t)n_gpus * 8 * sizeof(unsigned));
    }
    if ((n_gpus > 1 && !loopback) || force_rccl) {
        const std::string e = g_rccl.load();
        if (!e.empty()) return give_up(RT_ERR_HIP, e);
        m->comm.assign(n_gpus, nullptr);
        const ncclResult_t nr = g_rccl.CommInitAll(m->comm.data(), n_gpus, m->device.data());
        if (nr != ncclSuccess) { m->comm.clear(); return give_up(RT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr)); }
        m->use_rccl = true;
    }
    m->job_rc.assign(n_gpus, RT_OK);
    for (int r = 1; r < n_gpus; r++) m->workers.emplace_back(multi_worker, m, r);
    *out = m;
    return RT_OK;
}

void rt_multi_destroy(rt_multi_handle m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->quit = true;
    }
    m->cv_job.notify_all();
    for (auto& t : m->workers) t.join();
    for (size_t r = 0; r < m->stream.size(); r++) {
        hipSetDevice(m->device[r]);
        hipStreamSynchronize(m->stream[r]);
        hipStreamSynchronize(m->xstream[r]);
    }
    for (ncclComm_t c : m->comm) if (c) g_rccl.CommDestroy(c);
    multi_free_buffers(m);
    for (size_t r = 0; r < m->stream.size(); r++) {
        hipSetDevice(m->device[r]);
        hipStreamDestroy(m->stream[r]); hipStreamDestroy(m->xstream[r]);
    }
    for (int s = 0; s < MULTI_SLOTS; s++) {
        for (size_t r = 0; r < m->packed[s].size(); r++) {
            hipSetDevice(m->device[r]);
            hipEventDestroy(m->packed[s][r]); hipEventDestroy(m->bits_ready[s][r]); hipEventDestroy(m->sent[s][r]);
        }
        if (!m->device.empty()) hipSetDevice(m->device[0]);
        if (m->frame_done[s]) hipEventDestroy(m->frame_done[s]);
        if (m->h_ctrl[s]) hipHostFree(m->h_ctrl[s]);
    }
    for (rt_handle h : m->ctx) rt_destroy(h);
    delete m;
}

const char* rt_multi_last_error(rt_multi_handle m) { return m ? m->err.c_str() : g_multi_create_err.c_str(); }

int rt_multi_gpus(rt_multi_handle m) { return m ? m->n : RT_ERR_ARG; }

int rt_multi_uses_rccl(rt_multi_handle m) { return m ? (m->use_rccl ? 1 : 0) : RT_ERR_ARG; }

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
    if (m->frames_in_flight > 0) return mfail(m, RT_ERR_ARG, "rt_multi_upload_scene: a frame is still in flight (rt_multi_render_end first)");
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

int rt_multi_render_begin(rt_multi_handle m, int tile_rows) {
    if (!m) return RT_ERR_ARG;
    if (!m->have_scene) return mfail(m, RT_ERR_NO_SCENE, "rt_multi_render_begin: no scene uploaded");
    if (tile_rows < 0) return mfail(m, RT_ERR_ARG, "rt_multi_render_begin: bad argument");
    if (m->frames_in_flight >= MULTI_SLOTS) return mfail(m, RT_ERR_ARG, "rt_multi_render_begin: two frames are in flight already (rt_multi_render_end first)");
    return multi_begin(m, tile_rows);
}

int rt_multi_render_end(rt_multi_handle m, int32_t* img, uint64_t* rays) {
    if (!m) return RT_ERR_ARG;
    if (!img) return mfail(m, RT_ERR_ARG, "rt_multi_render_end: img is null");
    if (m->frames_in_flight == 0) return mfail(m, RT_ERR_ARG, "rt_multi_render_end: no frame in flight");
    return multi_end(m, img, rays);
}

int rt_multi_render_image(rt_multi_handle m, int tile_rows, int32_t* img, uint64_t* rays) {
    if (!m) return RT_ERR_ARG;
    if (!m->have_scene) return mfail(m, RT_ERR_NO_SCENE, "rt_multi_render_image: no scene uploaded");
    if (!img || tile_rows < 0) return mfail(m, RT_ERR_ARG, "rt_multi_render_image: bad argument");
    if (m->frames_in_flight > 0) return mfail(m, RT_ERR_ARG, "rt_multi_render_image: a frame begun with rt_multi_render_begin is still in flight");
    const int rc = multi_begin(m, tile_rows);
    if (rc != RT_OK) return rc;
    return multi_end(m, img, rays);
}

double rt_multi_last_ms(rt_multi_handle m) { return m ? m->last_ms : 0.0; }

int rt_multi_last_times(rt_multi_handle m, double* enqueue_ms, double* wait_ms) {
    if (!m) return RT_ERR_ARG;
    if (enqueue_ms) *enqueue_ms = m->last_enqueue_ms;
    if (wait_ms) *wait_ms = m->last_wait_ms;
    return RT_OK;
}

}  // extern "C"
