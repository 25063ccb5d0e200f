// Device check for tests/test_gpu_sqrt.py: r_sqrt(double) of crucible_amd/csrc/pathtrace.hpp (the wave-level short form of the
// compiler's f64 square root) against __builtin_sqrt, bit for bit, over random bit patterns and edge values.  A wave of
// ordinary numbers takes the short form, a wave that holds a zero, a subnormal, a tiny, an infinite, a negative or a NaN
// input takes the compiler's; both kinds are generated.  Prints "mismatches N of M"; exit code 1 when N > 0.
#include "pathtrace.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void sqrt_both(const double* x, double* fast, double* ref, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = cr::r_sqrt(x[i]);
    double v = x[i];
    asm volatile("" : "+v"(v));   // a second, independent evaluation
    ref[i] = __builtin_sqrt(v);
}

static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    const size_t n = (size_t)1 << 24;
    std::vector<double> x(n);
    uint64_t s = 12345;
    const double edges[] = {0.0, -0.0, 4.9e-324, 2.2250738585072014e-308, 0x1.0p-767, 0x1.fffffffffffffp-768, 0x1.0p-766, 1.0, 2.0, 0x1.fffffffffffffp1023,
                            __builtin_huge_val(), -1.0, __builtin_nan(""), 0x1.0p-1000, 0x1.0p1000, 3.0, 0.5, 0x1.0000000000001p0, 0x1.fffffffffffffp-1};
    for (size_t i = 0; i < n; i++) {
        uint64_t b = splitmix(s);
        const size_t wave = i / 64;
        if (wave % 4 == 0) b &= 0x7fffffffffffffffull;                                   // any non-negative pattern (tiny, subnormal, inf/NaN included)
        else if (wave % 4 == 1) b = (b & 0x000fffffffffffffull) | ((uint64_t)(0x100 + splitmix(s) % (0x7ff - 0x100)) << 52);   // ordinary: the short form
        else if (wave % 4 == 2) b = (b & 0x000fffffffffffffull) | ((uint64_t)(1023 - 40 + splitmix(s) % 80) << 52);               // around 1 (what the renderer sees)
        // else: any pattern at all
        memcpy(&x[i], &b, 8);
        if (wave % 64 == 63 && i % 64 < sizeof(edges) / sizeof(edges[0])) x[i] = edges[i % 64];
    }
    double *dx, *df, *dr;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&df, n * 8) != hipSuccess || hipMalloc(&dr, n * 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sqrt_both, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, df, dr, n);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
    std::vector<double> f(n), r(n);
    hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t a, b; memcpy(&a, &f[i], 8); memcpy(&b, &r[i], 8);
        const bool both_nan = f[i] != f[i] && r[i] != r[i];
        if (a != b && !both_nan) { if (bad < 5) fprintf(stderr, "x=%a fast=%a ref=%a\n", x[i], f[i], r[i]); bad++; }
    }
    printf("mismatches %zu of %zu\n", bad, n);
    return bad ? 1 : 0;
}
