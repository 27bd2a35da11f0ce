// expose(): bloom and 8-bit pack, for a whole frame and for the rows of one rank.
// A section of librt_hip.so's single translation unit: rt_device.hip includes the sections in dependency order.
#pragma once

namespace {

// ---- expose(): bloom + 8-bit pack (main.cpp:186-242, geometry.cpp:15-23) -------------------------
// Everything in double, every product and sum rounded on its own, taps in the reference's order (ky outer, kx
// inner), so the packed frame is the reference's bit for bit.  A tap outside the frame is skipped by the
// reference; here it reads +0.0 from the zero-filled border of the tile, and acc + (+0.0 * k) == acc because
// acc never is -0.0 (it starts at +0.0 and a sum of doubles is -0.0 only when both terms are).
constexpr int BLOOM_R = 20, BLOOM_K = 2 * BLOOM_R + 1, BLOOM_T = 16, BLOOM_TILE = BLOOM_T + 2 * BLOOM_R;

// B0 of main.cpp:198-205: exposure where the channel is over-exposed, else 0
__global__ void expose_mask_kernel(const double* rgb, double* cur, size_t n3, double exposure) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x)
        cur[i] = (rgb[i] * exposure > 1.0) ? exposure : 0.0;
}

// one pass of main.cpp:207-223 for one channel (blockIdx.z) of a 16x16 tile of pixels
// block_rows: null = every 16-row block of the frame (blockIdx.y is the block row); else the list of block rows to
// compute (a rank that blooms only what its own rows need, rt_expose_rows_device)
__global__ __launch_bounds__(BLOOM_T * BLOOM_T) void bloom_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                                  int W, int H, const double* __restrict__ kern,
                                                                  const int* __restrict__ block_rows) {
    __shared__ double tile[BLOOM_TILE][BLOOM_TILE + 1];
    const int ch = blockIdx.z;
    const int by = block_rows ? block_rows[blockIdx.y] : (int)blockIdx.y;
    const int x0 = blockIdx.x * BLOOM_T - BLOOM_R, y0 = by * BLOOM_T - BLOOM_R;
    bool lit = false;
    for (int i = threadIdx.x; i < BLOOM_TILE * BLOOM_TILE; i += BLOOM_T * BLOOM_T) {
        const int ty = i / BLOOM_TILE, tx = i - ty * BLOOM_TILE;
        const int x = x0 + tx, y = y0 + ty;
        const double v = (x >= 0 && x < W && y >= 0 && y < H) ? src[((size_t)y * W + x) * 3 + ch] : 0.0;
        tile[ty][tx] = v;
        lit |= (v != 0.0);
    }
    // a tile without a lit source pixel (most of a frame: only emitters and highlights saturate) sums 1681 products
    // +0.0 * k into +0.0; the barrier doubles as that test
    const bool any_lit = __syncthreads_or(lit ? 1 : 0) != 0;
    const int lx = threadIdx.x % BLOOM_T, ly = threadIdx.x / BLOOM_T;
    double c = 0.0;
    if (any_lit) {
        for (int ky = 0; ky < BLOOM_K; ky++) {
            const double* krow = kern + ky * BLOOM_K;
#pragma unroll
            for (int kx = 0; kx < BLOOM_K; kx++) c = c + tile[ly + ky][lx + kx] * krow[kx];
        }
    }
    const int x = blockIdx.x * BLOOM_T + lx, y = by * BLOOM_T + ly;
    if (x < W && y < H) dst[((size_t)y * W + x) * 3 + ch] = c;
}

// ---- expose() before the gather: every rank blooms and packs the rows it rendered -----------------------------
// B0 (main.cpp:198-205) is three bits per pixel.  Each rank reduces its rows to those bits (expose_bits_kernel), the
// ranks all-gather them (1 byte per pixel), and each rank then runs the blur passes on the 16-row blocks its own rows
// depend on: pass p of n needs pass p-1 within 20 rows, so the block set grows by two block rows per pass going
// backwards from the rank's own blocks (the halo).  Buffers are indexed by FRAME position, so a tap outside the frame is
// a tap outside the frame here too.  Every pixel goes through the same operations as in rt_expose_device.

// global row of a rank's packed row lr (cyclic row tiles)
DEV int tile_global_row(int lr, int tile_rows, int rank, int world) {
    return ((lr / tile_rows) * world + rank) * tile_rows + lr % tile_rows;
}

__global__ void expose_bits_kernel(const double* rgb, unsigned char* bits, size_t n, double exposure) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned b = 0;
        for (int k = 0; k < 3; k++) if (rgb[3 * i + k] * exposure > 1.0) b |= 1u << k;
        bits[i] = (unsigned char)b;
    }
}

// B0 of the listed block rows from the gathered bits: bits_all[r][lr][x] = rank r's packed row lr
__global__ void expose_b0_blocks_kernel(const unsigned char* __restrict__ bits_all, double* __restrict__ b0, int W, int H,
                                        int tile_rows, int world, int pad_rows, double exposure,
                                        const int* __restrict__ block_rows) {
    const int by = block_rows[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * BLOOM_T) return;
    const int y = by * BLOOM_T + t / W, x = t % W;
    if (y >= H) return;
    const int tile = y / tile_rows, r = tile % world, lr = (tile / world) * tile_rows + y % tile_rows;
    const unsigned b = bits_all[((size_t)r * pad_rows + lr) * W + x];
    double* o = b0 + ((size_t)y * W + x) * 3;
    for (int k = 0; k < 3; k++) o[k] = ((b >> k) & 1u) ? exposure : 0.0;
}

// expose_pack_kernel for a rank's packed rows; the bloom is read at the pixel's frame position
__global__ void expose_pack_rows_kernel(const double* rgb, const double* bloom, int* img, int W, int rows, int tile_rows,
                                        int rank, int world, double exposure) {
    const double mx = 1.0 / exposure;
    const size_t n = (size_t)rows * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lr = (int)(i / (size_t)W), x = (int)(i - (size_t)lr * W);
        const size_t f = (size_t)tile_global_row(lr, tile_rows, rank, world) * W + x;
        int v[3];
        for (int k = 0; k < 3; k++) {
            const double col = rgb[3 * i + k] + bloom[3 * f + k];
            int q = to_int_x86(255.0 * col / mx);
            if (q > 255) q = 255;
            v[k] = q;
        }
        img[i] = (int)(((unsigned)v[0] << 16) + ((unsigned)v[1] << 8) + (unsigned)v[2]);
    }
}

// main.cpp:225-227 + 237-241 with Color::to_int (geometry.cpp:15-23).  bloom == nullptr: no blur pass was
// asked for and the reference adds B0 itself.
__global__ void expose_pack_kernel(const double* rgb, const double* bloom, int* img, size_t n, double exposure) {
    const double mx = 1.0 / exposure;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int v[3];
        for (int k = 0; k < 3; k++) {
            const double c = rgb[3 * i + k];
            const double b = bloom ? bloom[3 * i + k] : ((c * exposure > 1.0) ? exposure : 0.0);
            const double col = c + b;
            int q = to_int_x86(255.0 * col / mx);
            if (q > 255) q = 255;
            v[k] = q;
        }
        img[i] = (int)(((unsigned)v[0] << 16) + ((unsigned)v[1] << 8) + (unsigned)v[2]);
    }
}

// fn: 0 sin, 1 cos, 2 acos, 3 pow, 4 x/y, 5 sqrt(x), 6 next RNG state of (uint64)x bits
__global__ void selftest_math_kernel(int fn, int n, const double* x, const double* y, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (fn) {
        case 0: r = rtm::sind(x[i]); break;
        case 1: r = rtm::cosd(x[i]); break;
        case 2: r = rtm::acosd(x[i]); break;
        case 3: r = rtm::powd(x[i], y[i]); break;
        case 4: r = x[i] / y[i]; break;
        case 5: r = __builtin_sqrt(x[i]); break;
        default: { unsigned s = (unsigned)__double_as_longlong(x[i]); rng_next(s); r = __longlong_as_double((long long)s); } break;
    }
    out[i] = r;
}

// broad-phase self-test: n independent (ray, record) pairs; flag[i] = 1 if the float test keeps the pair
__global__ void selftest_broad_kernel(int n, const double* o, const double* d, const float4* rec, float cmax,
                                      double tx, double ty, double tz, int* flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double T[3] = {tx, ty, tz};
    const RayRec r = broad_setup(dmk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dmk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), T, cmax);
    float hb, nq;
    const float sv = broad_s(rec[i], r.a, r.b, hb, nq);
    int f = 0;
    if (sv >= r.b.w) f = broad_behind(hb, nq, r.c) ? 2 : 1;   // 2 = dropped as "behind"
    flag[i] = f;
}

}  // namespace
