// Micro-benchmark: does the placement of a v_fma_f32 / v_fmac_f32's source registers over the four VGPR banks
// (register number mod 4) change its issue cost on gfx950?  32 independent instructions per loop body on 16
// accumulators, registers named in inline asm; 1..8 resident waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_vgpr_banks tools/ubench_vgpr_banks.hip && ./ubench_vgpr_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4000
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20"
template <int MODE>
__global__ void bench(float* out) {
    asm volatile("v_mov_b32 v20, 1.0\nv_mov_b32 v21, 0.5\nv_mov_b32 v22, 1.0\nv_mov_b32 v23, 0.5\nv_mov_b32 v24, 0.5\nv_mov_b32 v25, 1.0\nv_mov_b32 v26, 0.5\nv_mov_b32 v27, 1.0\n"
                 "v_mov_b32 v40, 0\nv_mov_b32 v41, 0\nv_mov_b32 v42, 0\nv_mov_b32 v43, 0\nv_mov_b32 v44, 0\nv_mov_b32 v45, 0\nv_mov_b32 v46, 0\nv_mov_b32 v47, 0\n"
                 "v_mov_b32 v48, 0\nv_mov_b32 v49, 0\nv_mov_b32 v50, 0\nv_mov_b32 v51, 0\nv_mov_b32 v52, 0\nv_mov_b32 v53, 0\nv_mov_b32 v54, 0\nv_mov_b32 v55, 0\ns_mov_b32 s20, 0.5\n" ::: CLOB);
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) asm volatile("v_fma_f32 v40, v21, v26, v40\nv_fma_f32 v41, v22, v27, v41\nv_fma_f32 v42, v23, v24, v42\nv_fma_f32 v43, v20, v25, v43\nv_fma_f32 v44, v21, v26, v44\nv_fma_f32 v45, v22, v27, v45\nv_fma_f32 v46, v23, v24, v46\nv_fma_f32 v47, v20, v25, v47\nv_fma_f32 v48, v21, v26, v48\nv_fma_f32 v49, v22, v27, v49\nv_fma_f32 v50, v23, v24, v50\nv_fma_f32 v51, v20, v25, v51\nv_fma_f32 v52, v21, v26, v52\nv_fma_f32 v53, v22, v27, v53\nv_fma_f32 v54, v23, v24, v54\nv_fma_f32 v55, v20, v25, v55\nv_fma_f32 v40, v21, v26, v40\nv_fma_f32 v41, v22, v27, v41\nv_fma_f32 v42, v23, v24, v42\nv_fma_f32 v43, v20, v25, v43\nv_fma_f32 v44, v21, v26, v44\nv_fma_f32 v45, v22, v27, v45\nv_fma_f32 v46, v23, v24, v46\nv_fma_f32 v47, v20, v25, v47\nv_fma_f32 v48, v21, v26, v48\nv_fma_f32 v49, v22, v27, v49\nv_fma_f32 v50, v23, v24, v50\nv_fma_f32 v51, v20, v25, v51\nv_fma_f32 v52, v21, v26, v52\nv_fma_f32 v53, v22, v27, v53\nv_fma_f32 v54, v23, v24, v54\nv_fma_f32 v55, v20, v25, v55" ::: CLOB);
        if (MODE == 1) asm volatile("v_fma_f32 v40, v21, v25, v40\nv_fma_f32 v41, v22, v26, v41\nv_fma_f32 v42, v23, v27, v42\nv_fma_f32 v43, v20, v24, v43\nv_fma_f32 v44, v21, v25, v44\nv_fma_f32 v45, v22, v26, v45\nv_fma_f32 v46, v23, v27, v46\nv_fma_f32 v47, v20, v24, v47\nv_fma_f32 v48, v21, v25, v48\nv_fma_f32 v49, v22, v26, v49\nv_fma_f32 v50, v23, v27, v50\nv_fma_f32 v51, v20, v24, v51\nv_fma_f32 v52, v21, v25, v52\nv_fma_f32 v53, v22, v26, v53\nv_fma_f32 v54, v23, v27, v54\nv_fma_f32 v55, v20, v24, v55\nv_fma_f32 v40, v21, v25, v40\nv_fma_f32 v41, v22, v26, v41\nv_fma_f32 v42, v23, v27, v42\nv_fma_f32 v43, v20, v24, v43\nv_fma_f32 v44, v21, v25, v44\nv_fma_f32 v45, v22, v26, v45\nv_fma_f32 v46, v23, v27, v46\nv_fma_f32 v47, v20, v24, v47\nv_fma_f32 v48, v21, v25, v48\nv_fma_f32 v49, v22, v26, v49\nv_fma_f32 v50, v23, v27, v50\nv_fma_f32 v51, v20, v24, v51\nv_fma_f32 v52, v21, v25, v52\nv_fma_f32 v53, v22, v26, v53\nv_fma_f32 v54, v23, v27, v54\nv_fma_f32 v55, v20, v24, v55" ::: CLOB);
        if (MODE == 2) asm volatile("v_fma_f32 v40, v20, v24, v40\nv_fma_f32 v41, v21, v25, v41\nv_fma_f32 v42, v22, v26, v42\nv_fma_f32 v43, v23, v27, v43\nv_fma_f32 v44, v20, v24, v44\nv_fma_f32 v45, v21, v25, v45\nv_fma_f32 v46, v22, v26, v46\nv_fma_f32 v47, v23, v27, v47\nv_fma_f32 v48, v20, v24, v48\nv_fma_f32 v49, v21, v25, v49\nv_fma_f32 v50, v22, v26, v50\nv_fma_f32 v51, v23, v27, v51\nv_fma_f32 v52, v20, v24, v52\nv_fma_f32 v53, v21, v25, v53\nv_fma_f32 v54, v22, v26, v54\nv_fma_f32 v55, v23, v27, v55\nv_fma_f32 v40, v20, v24, v40\nv_fma_f32 v41, v21, v25, v41\nv_fma_f32 v42, v22, v26, v42\nv_fma_f32 v43, v23, v27, v43\nv_fma_f32 v44, v20, v24, v44\nv_fma_f32 v45, v21, v25, v45\nv_fma_f32 v46, v22, v26, v46\nv_fma_f32 v47, v23, v27, v47\nv_fma_f32 v48, v20, v24, v48\nv_fma_f32 v49, v21, v25, v49\nv_fma_f32 v50, v22, v26, v50\nv_fma_f32 v51, v23, v27, v51\nv_fma_f32 v52, v20, v24, v52\nv_fma_f32 v53, v21, v25, v53\nv_fma_f32 v54, v22, v26, v54\nv_fma_f32 v55, v23, v27, v55" ::: CLOB);
        if (MODE == 3) asm volatile("v_fmac_f32_e32 v40, v21, v26\nv_fmac_f32_e32 v41, v22, v27\nv_fmac_f32_e32 v42, v23, v24\nv_fmac_f32_e32 v43, v20, v25\nv_fmac_f32_e32 v44, v21, v26\nv_fmac_f32_e32 v45, v22, v27\nv_fmac_f32_e32 v46, v23, v24\nv_fmac_f32_e32 v47, v20, v25\nv_fmac_f32_e32 v48, v21, v26\nv_fmac_f32_e32 v49, v22, v27\nv_fmac_f32_e32 v50, v23, v24\nv_fmac_f32_e32 v51, v20, v25\nv_fmac_f32_e32 v52, v21, v26\nv_fmac_f32_e32 v53, v22, v27\nv_fmac_f32_e32 v54, v23, v24\nv_fmac_f32_e32 v55, v20, v25\nv_fmac_f32_e32 v40, v21, v26\nv_fmac_f32_e32 v41, v22, v27\nv_fmac_f32_e32 v42, v23, v24\nv_fmac_f32_e32 v43, v20, v25\nv_fmac_f32_e32 v44, v21, v26\nv_fmac_f32_e32 v45, v22, v27\nv_fmac_f32_e32 v46, v23, v24\nv_fmac_f32_e32 v47, v20, v25\nv_fmac_f32_e32 v48, v21, v26\nv_fmac_f32_e32 v49, v22, v27\nv_fmac_f32_e32 v50, v23, v24\nv_fmac_f32_e32 v51, v20, v25\nv_fmac_f32_e32 v52, v21, v26\nv_fmac_f32_e32 v53, v22, v27\nv_fmac_f32_e32 v54, v23, v24\nv_fmac_f32_e32 v55, v20, v25" ::: CLOB);
        if (MODE == 4) asm volatile("v_fmac_f32_e32 v40, v21, v25\nv_fmac_f32_e32 v41, v22, v26\nv_fmac_f32_e32 v42, v23, v27\nv_fmac_f32_e32 v43, v20, v24\nv_fmac_f32_e32 v44, v21, v25\nv_fmac_f32_e32 v45, v22, v26\nv_fmac_f32_e32 v46, v23, v27\nv_fmac_f32_e32 v47, v20, v24\nv_fmac_f32_e32 v48, v21, v25\nv_fmac_f32_e32 v49, v22, v26\nv_fmac_f32_e32 v50, v23, v27\nv_fmac_f32_e32 v51, v20, v24\nv_fmac_f32_e32 v52, v21, v25\nv_fmac_f32_e32 v53, v22, v26\nv_fmac_f32_e32 v54, v23, v27\nv_fmac_f32_e32 v55, v20, v24\nv_fmac_f32_e32 v40, v21, v25\nv_fmac_f32_e32 v41, v22, v26\nv_fmac_f32_e32 v42, v23, v27\nv_fmac_f32_e32 v43, v20, v24\nv_fmac_f32_e32 v44, v21, v25\nv_fmac_f32_e32 v45, v22, v26\nv_fmac_f32_e32 v46, v23, v27\nv_fmac_f32_e32 v47, v20, v24\nv_fmac_f32_e32 v48, v21, v25\nv_fmac_f32_e32 v49, v22, v26\nv_fmac_f32_e32 v50, v23, v27\nv_fmac_f32_e32 v51, v20, v24\nv_fmac_f32_e32 v52, v21, v25\nv_fmac_f32_e32 v53, v22, v26\nv_fmac_f32_e32 v54, v23, v27\nv_fmac_f32_e32 v55, v20, v24" ::: CLOB);
        if (MODE == 5) asm volatile("v_fmac_f32_e32 v40, v20, v24\nv_fmac_f32_e32 v41, v21, v25\nv_fmac_f32_e32 v42, v22, v26\nv_fmac_f32_e32 v43, v23, v27\nv_fmac_f32_e32 v44, v20, v24\nv_fmac_f32_e32 v45, v21, v25\nv_fmac_f32_e32 v46, v22, v26\nv_fmac_f32_e32 v47, v23, v27\nv_fmac_f32_e32 v48, v20, v24\nv_fmac_f32_e32 v49, v21, v25\nv_fmac_f32_e32 v50, v22, v26\nv_fmac_f32_e32 v51, v23, v27\nv_fmac_f32_e32 v52, v20, v24\nv_fmac_f32_e32 v53, v21, v25\nv_fmac_f32_e32 v54, v22, v26\nv_fmac_f32_e32 v55, v23, v27\nv_fmac_f32_e32 v40, v20, v24\nv_fmac_f32_e32 v41, v21, v25\nv_fmac_f32_e32 v42, v22, v26\nv_fmac_f32_e32 v43, v23, v27\nv_fmac_f32_e32 v44, v20, v24\nv_fmac_f32_e32 v45, v21, v25\nv_fmac_f32_e32 v46, v22, v26\nv_fmac_f32_e32 v47, v23, v27\nv_fmac_f32_e32 v48, v20, v24\nv_fmac_f32_e32 v49, v21, v25\nv_fmac_f32_e32 v50, v22, v26\nv_fmac_f32_e32 v51, v23, v27\nv_fmac_f32_e32 v52, v20, v24\nv_fmac_f32_e32 v53, v21, v25\nv_fmac_f32_e32 v54, v22, v26\nv_fmac_f32_e32 v55, v23, v27" ::: CLOB);
        if (MODE == 6) asm volatile("v_fmac_f32_e32 v40, s20, v26\nv_fmac_f32_e32 v41, s20, v27\nv_fmac_f32_e32 v42, s20, v24\nv_fmac_f32_e32 v43, s20, v25\nv_fmac_f32_e32 v44, s20, v26\nv_fmac_f32_e32 v45, s20, v27\nv_fmac_f32_e32 v46, s20, v24\nv_fmac_f32_e32 v47, s20, v25\nv_fmac_f32_e32 v48, s20, v26\nv_fmac_f32_e32 v49, s20, v27\nv_fmac_f32_e32 v50, s20, v24\nv_fmac_f32_e32 v51, s20, v25\nv_fmac_f32_e32 v52, s20, v26\nv_fmac_f32_e32 v53, s20, v27\nv_fmac_f32_e32 v54, s20, v24\nv_fmac_f32_e32 v55, s20, v25\nv_fmac_f32_e32 v40, s20, v26\nv_fmac_f32_e32 v41, s20, v27\nv_fmac_f32_e32 v42, s20, v24\nv_fmac_f32_e32 v43, s20, v25\nv_fmac_f32_e32 v44, s20, v26\nv_fmac_f32_e32 v45, s20, v27\nv_fmac_f32_e32 v46, s20, v24\nv_fmac_f32_e32 v47, s20, v25\nv_fmac_f32_e32 v48, s20, v26\nv_fmac_f32_e32 v49, s20, v27\nv_fmac_f32_e32 v50, s20, v24\nv_fmac_f32_e32 v51, s20, v25\nv_fmac_f32_e32 v52, s20, v26\nv_fmac_f32_e32 v53, s20, v27\nv_fmac_f32_e32 v54, s20, v24\nv_fmac_f32_e32 v55, s20, v25" ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v40, v47\nv_add_f32 %0, %0, v55" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char* name) {
    printf("%-46s", name);
    for (int waves : {1, 2, 4, 8}) {
        const int bs = 256 * (waves > 4 ? 4 : waves), per_cu = waves > 4 ? 2 : 1, grid = 256 * per_cu;
        float* out; hipMalloc(&out, (size_t)grid * bs * 4);
        bench<MODE><<<grid, bs>>>(out); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); bench<MODE><<<grid, bs>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  w%d: %5.2f ns*SIMD/instr", waves, ms * 1e6 / ((double)ITERS * 32 * waves));
        hipFree(out);
    }
    printf("\n");
}
int main() {
    run<0>("v_fma_f32, sources on three banks");
    run<1>("v_fma_f32, both factors on one bank");
    run<2>("v_fma_f32, all three on one bank");
    run<3>("v_fmac_f32, three banks");
    run<4>("v_fmac_f32, both factors on one bank");
    run<5>("v_fmac_f32, all on one bank");
    run<6>("v_fmac_f32, one factor an SGPR");
    return 0;
}
