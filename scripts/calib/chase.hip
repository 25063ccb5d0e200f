// Calibration: what a CU sustains on the access pattern of a BVH walk through global memory -- every lane follows its
// own chain of records (the next index comes out of the record just read), R bytes per record read with 16-B loads,
// W waves per SIMD resident, table of T bytes (L1-, L2-, Infinity-Cache- or HBM-resident).
//   hipcc --offload-arch=gfx950 -O3 -o chase chase.hip && ./chase [out.json]
// Output: wave-steps per 1000 shader cycles per CU (one wave-step = 64 lanes each reading one record) and the latency
// of one dependent step seen by a single wave.  The walk of the 1M-sphere scene (f64 wrappers = 64-B records, 16 waves
// per CU) can be held against the 64-B rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// A record is R/16 vectors of 4 words; word 0 of vector 0 holds the next record's index (a random permutation cycle
// per table), the other vectors are folded in so that every load is used.
template <int R>
__global__ void __launch_bounds__(1024) chase_kernel(const u4* __restrict__ table, uint32_t n_records, int steps, uint32_t* sink,
                                                     unsigned long long* cycles) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_records;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s++) {
        const u4* rec = table + (size_t)idx * (R / 16);
        u4 v0 = rec[0];
        uint32_t fold = 0;
#pragma unroll
        for (int k = 1; k < R / 16; k++) { u4 v = rec[k]; fold ^= v.x ^ v.y ^ v.z ^ v.w; }
        acc += fold;
        idx = (v0.x + (fold & 0u)) % n_records;   // the next index waits for every load of the record
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) *sink = idx;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 1) sink[1 + blockIdx.x] = idx;
}

template <int R>
int run(const char* label, size_t table_bytes, int waves_per_simd, int n_cus, FILE* json, bool first) {
    const uint32_t n = (uint32_t)(table_bytes / R);
    std::vector<uint32_t> host((size_t)n * (R / 4));
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; i++) perm[i] = i;
    std::mt19937 rng(12345);
    for (uint32_t i = n - 1; i > 0; i--) { uint32_t j = rng() % (i + 1); std::swap(perm[i], perm[j]); }
    for (uint32_t i = 0; i < n; i++) {   // one cycle through all records: record perm[i] points to perm[i + 1]
        uint32_t* r = &host[(size_t)perm[i] * (R / 4)];
        for (int k = 1; k < R / 4; k++) r[k] = (uint32_t)i * 2654435761u + (uint32_t)k;
        r[0] = perm[(i + 1) % n];
    }
    u4* d; uint32_t* sink; unsigned long long* cyc;
    const int block = 256 * waves_per_simd;   // one workgroup per CU: 4 SIMDs x waves_per_simd waves
    CHECK(hipMalloc(&d, host.size() * 4)); CHECK(hipMalloc(&sink, (2 + n_cus) * 4)); CHECK(hipMalloc(&cyc, n_cus * 8));
    CHECK(hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    const int steps = 2000;
    double best = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL((chase_kernel<R>), dim3(n_cus), dim3(block), 0, 0, d, n, steps, sink, cyc);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> c(n_cus);
        CHECK(hipMemcpy(c.data(), cyc, n_cus * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto x : c) sum += (double)x;
        if (rep > 0) best = std::min(best, sum / n_cus);
    }
    // __builtin_readcyclecounter ticks at the 100 MHz reference on this target: convert with the measured shader clock
    const double cycles_per_step = best / steps;   // in counter ticks
    const int waves = 4 * waves_per_simd;
    printf("%-10s R=%3d B  %2d waves/CU  %10.1f ticks per dependent step  -> %8.3f wave-steps per tick per CU\n", label, R, waves, cycles_per_step, waves / cycles_per_step);
    if (json) fprintf(json, "%s{\"table\": \"%s\", \"record_bytes\": %d, \"waves_per_cu\": %d, \"ticks_per_step\": %.2f, \"wave_steps_per_tick_per_cu\": %.4f}", first ? "" : ", ", label, R, waves, cycles_per_step, waves / cycles_per_step);
    CHECK(hipFree(d)); CHECK(hipFree(sink)); CHECK(hipFree(cyc));
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    FILE* json = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (json) fprintf(json, "{\"device\": \"%s\", \"cus\": %d, \"tick\": \"__builtin_readcyclecounter (s_memtime)\", \"rows\": [", prop.gcnArchName, n_cus);
    bool first = true;
    if (const char* only = getenv("CHASE_ONLY")) {   // "R,table_MiB,waves_per_simd": one configuration (for a counter pass)
        int R = 0, mb = 0, w = 0;
        if (sscanf(only, "%d,%d,%d", &R, &mb, &w) != 3) { fprintf(stderr, "CHASE_ONLY=R,table_MiB,waves_per_simd\n"); return 2; }
        const size_t bytes = (size_t)mb << 20;
        // per launch: n_cus workgroups x (256 * w) lanes x 2000 steps, one R-byte record each; 4 launches
        printf("known bytes per launch: %.0f (lanes %d x steps 2000 x %d B)\n", (double)n_cus * 256 * w * 2000.0 * R, n_cus * 256 * w, R);
        int rc = R == 16 ? run<16>("only", bytes, w, n_cus, json, true) : R == 32 ? run<32>("only", bytes, w, n_cus, json, true)
               : R == 64 ? run<64>("only", bytes, w, n_cus, json, true) : run<96>("only", bytes, w, n_cus, json, true);
        if (json) { fprintf(json, "]}\n"); fclose(json); }
        return rc;
    }
    struct T { const char* label; size_t bytes; } tables[] = {{"L1 16KB", 16u << 10}, {"L2 2MB", 2u << 20}, {"MALL 64MB", 64u << 20}, {"HBM 512MB", 512u << 20}};
    for (const T& t : tables)
        for (int w : {1, 2, 4}) {
            if (run<16>(t.label, t.bytes, w, n_cus, json, first)) return 1;
            first = false;
            if (run<32>(t.label, t.bytes, w, n_cus, json, first)) return 1;
            if (run<64>(t.label, t.bytes, w, n_cus, json, first)) return 1;
            if (w == 4 && run<96>(t.label, t.bytes, w, n_cus, json, first)) return 1;
        }
    if (json) { fprintf(json, "]}\n"); fclose(json); }
    return 0;
}
