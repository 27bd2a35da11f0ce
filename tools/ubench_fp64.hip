// Micro-benchmark: issue cost of double-precision VALU instruction patterns on one SIMD of gfx950,
// for 1..8 resident waves per SIMD and for dependent vs independent chains.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o ubench_fp64 tools/ubench_fp64.hip && ./ubench_fp64
// Prints cycles per wave-instruction per SIMD (shader clock, s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define ITERS 2000

template <int MODE>
__global__ void bench(double* out, long long* cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0;
    double a4 = a0 + 4.0, a5 = a0 + 5.0, a6 = a0 + 6.0, a7 = a0 + 7.0;
    const double m = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) {          // 8 dependent fma on one chain
#pragma unroll
            for (int k = 0; k < 8; k++) a0 = __builtin_fma(a0, m, c);
        } else if (MODE == 1) {   // 8 independent fma (8 chains)
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        } else if (MODE == 2) {   // 8 dependent mul
#pragma unroll
            for (int k = 0; k < 8; k++) a0 = a0 * m;
        } else if (MODE == 3) {   // 8 dependent add
#pragma unroll
            for (int k = 0; k < 8; k++) a0 = a0 + c;
        } else if (MODE == 4) {   // 8 independent divisions
            a0 = a0 / m; a1 = a1 / m; a2 = a2 / m; a3 = a3 / m; a4 = a4 / m; a5 = a5 / m; a6 = a6 / m; a7 = a7 / m;
        } else if (MODE == 5) {   // 8 dependent divisions
#pragma unroll
            for (int k = 0; k < 8; k++) a0 = m / a0;
        } else if (MODE == 6) {   // 8 dependent sqrt
#pragma unroll
            for (int k = 0; k < 8; k++) a0 = __builtin_sqrt(a0 + 2.0);
        } else if (MODE == 7) {   // 8 independent sqrt
            a0 = __builtin_sqrt(a0 + 2.0); a1 = __builtin_sqrt(a1 + 2.0); a2 = __builtin_sqrt(a2 + 2.0); a3 = __builtin_sqrt(a3 + 2.0);
            a4 = __builtin_sqrt(a4 + 2.0); a5 = __builtin_sqrt(a5 + 2.0); a6 = __builtin_sqrt(a6 + 2.0); a7 = __builtin_sqrt(a7 + 2.0);
        } else if (MODE == 8) {   // 8 dependent f32 fma
#pragma unroll
            for (int k = 0; k < 8; k++) f0 = __builtin_fmaf(f0, 1.0000001f, 1e-9f);
        } else if (MODE == 9) {   // 8 independent f32 fma (4 chains x 2)
            f0 = __builtin_fmaf(f0, 1.0000001f, 1e-9f); f1 = __builtin_fmaf(f1, 1.0000001f, 1e-9f);
            f2 = __builtin_fmaf(f2, 1.0000001f, 1e-9f); f3 = __builtin_fmaf(f3, 1.0000001f, 1e-9f);
            f0 = __builtin_fmaf(f0, 1.0000001f, 1e-9f); f1 = __builtin_fmaf(f1, 1.0000001f, 1e-9f);
            f2 = __builtin_fmaf(f2, 1.0000001f, 1e-9f); f3 = __builtin_fmaf(f3, 1.0000001f, 1e-9f);
        } else if (MODE == 10) {  // 8 dependent v_rcp_f64
#pragma unroll
            for (int k = 0; k < 8; k++) a0 = __builtin_amdgcn_rcp(a0);
        } else if (MODE == 11) {  // 4 dependent fma interleaved with 4 dependent fma of another chain
#pragma unroll
            for (int k = 0; k < 4; k++) { a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per8) {
    int dev_cus = 256;
    printf("%-34s", name);
    for (int waves : {1, 2, 4, 8}) {
        const int block = 64 * 4 * waves;     // one workgroup per CU: `waves` waves on each of its 4 SIMDs
        if (block > 1024) {
            // two workgroups of 1024 per CU
        }
        const int blocks_per_cu = block > 1024 ? block / 1024 : 1;
        const int bs = block > 1024 ? 1024 : block;
        const int grid = dev_cus * blocks_per_cu;
        double* out; long long* cyc;
        hipMalloc(&out, (size_t)grid * bs * 8);
        hipMalloc(&cyc, (size_t)grid * bs / 64 * 8);
        bench<MODE><<<grid, bs>>>(out, cyc, 1.5);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        bench<MODE><<<grid, bs>>>(out, cyc, 1.5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h((size_t)grid * bs / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        // s_memtime ticks at 100 MHz on this part? report both: ticks per instr and wall ns per instr per SIMD
        const double instr = (double)ITERS * per8;
        printf("  w%d: %7.2f tick/instr/wave %6.2f ns*SIMD/instr", waves, avg / instr, ms * 1e6 / (instr * waves));
        hipFree(out); hipFree(cyc);
    }
    printf("\n");
}

int main() {
    run<0>("fma f64 dependent", 8);
    run<1>("fma f64 independent x8", 8);
    run<11>("fma f64 2 chains", 8);
    run<2>("mul f64 dependent", 8);
    run<3>("add f64 dependent", 8);
    run<4>("div f64 independent (per div)", 8);
    run<5>("div f64 dependent (per div)", 8);
    run<6>("sqrt f64 dependent (per sqrt+add)", 8);
    run<7>("sqrt f64 independent", 8);
    run<8>("fma f32 dependent", 8);
    run<9>("fma f32 4 chains", 8);
    run<10>("v_rcp_f64 dependent", 8);
    return 0;
}
