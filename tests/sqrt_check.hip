// Device check for tests/test_gpu_sqrt.py: r_sqrt(double) of crucible_amd/csrc/pathtrace.hpp (the wave-level short form of the
// compiler's f64 square root) against __builtin_sqrt, bit for bit, over random bit patterns and edge values; and as_i32 (the
// conversion instruction) against its guarded form as_i32_reference, f64 and f32.  A wave of
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

__global__ void cvt_both(const double* x, int32_t* fast, int32_t* ref, int32_t* fast32, int32_t* ref32, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = cr::as_i32(x[i]);
    ref[i] = cr::as_i32_reference(x[i]);
    const float f = (float)x[i];
    fast32[i] = cr::as_i32(f);
    ref32[i] = cr::as_i32_reference(f);
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
    // `as i32` (as_i32 against its guarded form), f64 and f32: the same inputs plus values around the i32 range
    const double iedges[] = {2147483647.0, 2147483648.0, 2147483646.5, 2147483647.5, -2147483648.0, -2147483649.0, -2147483648.5, -2147483647.5,
                             4294967296.0, -4294967296.0, 1e300, -1e300, 0.5, -0.5, 0.999999, -0.999999, 1.5, -1.5, 2.5, -2.5, 1e9, -1e9};
    for (size_t i = 0; i < n; i++) {
        const size_t wave = i / 64;
        if (wave % 4 == 3) { const uint64_t b = splitmix(s); x[i] = ((double)(int64_t)b) * 0x1.0p-31; }   // spread over about +-2^32
        if (wave % 64 == 61 && i % 64 < sizeof(iedges) / sizeof(iedges[0])) x[i] = iedges[i % 64];
    }
    int32_t *di, *dj, *dk, *dl;
    if (hipMalloc(&di, n * 4) != hipSuccess || hipMalloc(&dj, n * 4) != hipSuccess || hipMalloc(&dk, n * 4) != hipSuccess || hipMalloc(&dl, n * 4) != hipSuccess) return 2;
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_both, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, di, dj, dk, dl, n);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
    std::vector<int32_t> a(n), b(n), c(n), d(n);
    hipMemcpy(a.data(), di, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dj, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dk, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dl, n * 4, hipMemcpyDeviceToHost);
    size_t bad_cvt = 0;
    for (size_t i = 0; i < n; i++) if (a[i] != b[i] || c[i] != d[i]) { if (bad_cvt < 5) fprintf(stderr, "x=%a as_i32 %d / %d, f32 %d / %d\n", x[i], a[i], b[i], c[i], d[i]); bad_cvt++; }
    printf("as_i32 mismatches %zu of %zu\n", bad_cvt, n);
    return (bad || bad_cvt) ? 1 : 0;
}
