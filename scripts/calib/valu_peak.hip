// Calibration: issue rate of independent wave64 v_fma_f32 / v_pk_fma-free code at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) fma_loop(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.000001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(float));
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int iters = 1 << 15;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int block = 256, blocks_per_cu = waves_per_simd;   // 256 threads = 4 waves = 1 wave per SIMD
        int grid = p.multiProcessorCount * blocks_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        fma_loop<<<grid, block>>>(d, 64); hipDeviceSynchronize();
        hipEventRecord(e0); fma_loop<<<grid, block>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr = (double)grid * (block / 64) * iters * 8.0;
        double per_simd_per_s = wave_instr / (p.multiProcessorCount * 4) / (ms * 1e-3);
        printf("waves/SIMD %d: %.3f ms, %.2f G wave-FMA/s per SIMD (clock ~2.4 GHz => %.2f cycles per wave-instr), %.1f TFLOP/s\n",
               waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 128 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
