// Calibration: what this chip's HBM sustains, to stand beside the 8 TB/s figure the roofline divides by (SURVEY 8(d):
// "confirm on the box with a hipMemcpyDtoD / stream-triad probe and report the measured peak alongside").
//   hipcc --offload-arch=gfx950 -O3 -o hbm_peak hbm_peak.hip && ./hbm_peak [out.json]
// Four kernels over arrays far larger than the 256 MB Infinity Cache (2 GiB each), 16 B per lane per access, grid-stride:
//   read   : sum of one array                     bytes = N
//   write  : fill of one array                    bytes = N
//   copy   : a[i] = b[i]                          bytes = 2N   (and hipMemcpyDtoD of the same arrays)
//   triad  : a[i] = b[i] + s * c[i]               bytes = 3N
// Best of 5 timed launches each, HIP events on the launch stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ b, size_t n, float* sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += b[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;   // keeps the loads
}
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ a, size_t n, float v) {
    const f4 x = {v, v, v, v};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = x;
}
__global__ void __launch_bounds__(256) k_copy(f4* __restrict__ a, const f4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i];
}
__global__ void __launch_bounds__(256) k_triad(f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, size_t n, float s) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i] + s * c[i];
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(f4);
    f4 *a, *b, *c;
    float* sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[5] = {"read", "write", "copy", "triad", "hipMemcpyDtoD"};
    const double traffic[5] = {1.0, 1.0, 2.0, 3.0, 2.0};
    double best_gbps[5] = {0, 0, 0, 0, 0};
    int best_grid[5] = {0, 0, 0, 0, 0};
    for (int blocks_per_cu : {4, 8, 16, 32}) {
        const unsigned grid = (unsigned)(prop.multiProcessorCount * blocks_per_cu);
        for (int k = 0; k < 5; k++) {
            if (k == 4 && blocks_per_cu != 4) continue;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipEventRecord(e0, st));
                if (k == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, b, n, sink);
                else if (k == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, st, a, n, 1.0f);
                else if (k == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, a, b, n);
                else if (k == 3) hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, st, a, b, c, n, 0.5f);
                else CHECK(hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, st));
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 0) continue;   // warm-up
                const double gbps = traffic[k] * (double)bytes / (ms * 1e-3) / 1e9;
                if (gbps > best_gbps[k]) { best_gbps[k] = gbps; best_grid[k] = (int)grid; }
            }
        }
    }
    printf("%s %s, %d CUs; 2 GiB arrays, best of 5 per grid size\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    for (int k = 0; k < 5; k++) printf("%-14s %8.1f GB/s  (grid %d x 256)\n", names[k], best_gbps[k], best_grid[k]);
    const double peak = *std::max_element(best_gbps, best_gbps + 5);
    printf("measured peak  %8.1f GB/s = %.2f of the 8000 GB/s the roofline uses\n", peak, peak / 8000.0);
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (!f) { perror(argv[1]); return 1; }
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"array_bytes\": %zu", prop.gcnArchName, prop.multiProcessorCount, bytes);
        for (int k = 0; k < 5; k++) fprintf(f, ", \"%s_GBps\": %.1f", names[k], best_gbps[k]);
        fprintf(f, ", \"measured_peak_GBps\": %.1f, \"nominal_peak_GBps\": 8000.0}\n", peak);
        fclose(f);
    }
    return 0;
}
