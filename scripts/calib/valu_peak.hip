// Calibration: VALU issue rates on this chip, in shader cycles (s_memtime), per instruction kind and per number of
// resident waves per SIMD.  Every kernel runs 8 independent dependency chains of one instruction per lane, so a single
// wave already has 8 instructions in flight; the clock is read inside the kernel, so DVFS does not enter the result.
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
// Output: cycles per wave64 instruction per SIMD (1 / wave-instr per SIMD-cycle); the guide's figure for v_fma_f32 is 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHAIN8(OP)                                                                                              \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

enum Kind { FMA_F32, PK_FMA_F32, ADD_F32, MAX_F32, MAX3_F32, CNDMASK, ADD_U32, ADD_F64, MUL_F64, FMA_F64, MAX_F64, RCP_F64, SQRT_F32,
            CND_VCC_SET, CND_SGPR, CND_SGPR_ALT, MIN_F32, PK_ADD_F32, PK_MUL_F32, CND_E64_VCC, CMP_CND_VCC, CMP_CND_SGPR, CMP_VCC, CMP_SGPR, CMP_CND2_VCC, CMP_CND2_E64, N_KINDS };
static const char* kind_name[N_KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_max_f32", "v_max3_f32", "v_cndmask_b32", "v_add_u32",
                                         "v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_rcp_f64", "v_sqrt_f32",
                                         "v_cndmask_b32(vcc written by v_cmp)", "v_cndmask_b32_e64(sgpr pair)", "v_cndmask_b32_e64(two sgpr pairs)", "v_min_f32", "v_pk_add_f32", "v_pk_mul_f32",
                                         "v_cndmask_b32_e64(vcc)", "v_cmp_e32 vcc + v_cndmask_e32 (per pair)", "v_cmp_e64 sgpr + v_cndmask_e64 (per pair)", "v_cmp_lt_f32_e32 (vcc)", "v_cmp_lt_f32_e64 (sgpr pair)",
                                         "v_cmp_e32 + 2 x v_cndmask_e32, an f64 select as compiled (per triple)", "v_cmp_e32 + 2 x v_cndmask_e64 vcc (per triple)"};

template <int KIND>
__global__ void __launch_bounds__(256) loop_kernel(unsigned long long* cycles, float* sink, int iters) {
    float f[8]; double d[8]; unsigned u[8], u2[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x + i; d[i] = threadIdx.x + i + 0.5; u[i] = threadIdx.x + i; u2[i] = threadIdx.x + 2 * i; p[i] = f2{f[i], f[i] + 1}; }
    const float b = 1.000001f, c = 0.5f;
    const double bd = 1.000001, cd = 0.5;
    const f2 bp = {b, b}, cp = {c, c};
    const unsigned tid = threadIdx.x;
    unsigned long long m0 = 0x5555555555555555ull, m1 = 0x0f0f0f0f0f0f0f0full;
    asm volatile("s_mov_b64 %0, %0" : "+s"(m0));
    asm volatile("s_mov_b64 %0, %0" : "+s"(m1));
    if (KIND == CND_VCC_SET) asm volatile("v_cmp_gt_u32 vcc, 32, %0" :: "v"(tid) : "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define OP_FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(c));
#define OP_PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(bp), "v"(cp));
#define OP_ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
#define OP_MAX_F32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
#define OP_MAX3_F32(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c), "v"(b));
#define OP_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(tid));
// Round 3: the 12.6-cycle v_cndmask_b32 of round 2 read a VCC that nothing in the kernel had ever written.  The variants below
// select on a mask a v_cmp wrote first (VCC), on an SGPR pair, and alternately on two SGPR pairs.
#define OP_CND_E64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(tid), "s"(m0));
#define OP_CND_E64_ALT(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(tid), "s"((i & 1) ? m1 : m0));
#define OP_CND_E64_VCC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(tid));
#define OP_CMP_CND_VCC(i) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(f[i]), "v"(c), "v"(tid) : "vcc");
#define OP_CMP_CND_SGPR(i) asm volatile("v_cmp_lt_f32_e64 %4, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, %4" : "+v"(u[i]) : "v"(f[i]), "v"(c), "v"(tid), "s"(m0));
#define OP_CMP_VCC(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(f[i]), "v"(c) : "vcc");
#define OP_CMP_SGPR(i) asm volatile("v_cmp_lt_f32_e64 %2, %0, %1" :: "v"(f[i]), "v"(c), "s"(m0));
#define OP_CMP_CND2_VCC(i) asm volatile("v_cmp_lt_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc" : "+v"(u[i]), "+v"(u2[i]) : "v"(f[i]), "v"(c), "v"(tid) : "vcc");
#define OP_CMP_CND2_E64(i) asm volatile("v_cmp_lt_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc" : "+v"(u[i]), "+v"(u2[i]) : "v"(f[i]), "v"(c), "v"(tid) : "vcc");
#define OP_MIN_F32(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
#define OP_PK_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
#define OP_PK_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bp));
#define OP_ADD_U32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define OP_ADD_F64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define OP_MUL_F64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
#define OP_FMA_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(bd), "v"(cd));
#define OP_MAX_F64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define OP_RCP_F64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define OP_SQRT_F32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
        for (int r = 0; r < 4; r++) {
            if (KIND == FMA_F32) { CHAIN8(OP_FMA_F32) }
            else if (KIND == PK_FMA_F32) { CHAIN8(OP_PK) }
            else if (KIND == ADD_F32) { CHAIN8(OP_ADD_F32) }
            else if (KIND == MAX_F32) { CHAIN8(OP_MAX_F32) }
            else if (KIND == MAX3_F32) { CHAIN8(OP_MAX3_F32) }
            else if (KIND == CNDMASK) { CHAIN8(OP_CND) }
            else if (KIND == ADD_U32) { CHAIN8(OP_ADD_U32) }
            else if (KIND == ADD_F64) { CHAIN8(OP_ADD_F64) }
            else if (KIND == MUL_F64) { CHAIN8(OP_MUL_F64) }
            else if (KIND == FMA_F64) { CHAIN8(OP_FMA_F64) }
            else if (KIND == MAX_F64) { CHAIN8(OP_MAX_F64) }
            else if (KIND == RCP_F64) { CHAIN8(OP_RCP_F64) }
            else if (KIND == CND_VCC_SET) { CHAIN8(OP_CND) }
            else if (KIND == CND_SGPR) { CHAIN8(OP_CND_E64) }
            else if (KIND == CND_SGPR_ALT) { CHAIN8(OP_CND_E64_ALT) }
            else if (KIND == MIN_F32) { CHAIN8(OP_MIN_F32) }
            else if (KIND == CND_E64_VCC) { CHAIN8(OP_CND_E64_VCC) }
            else if (KIND == CMP_CND_VCC) { CHAIN8(OP_CMP_CND_VCC) }
            else if (KIND == CMP_CND_SGPR) { CHAIN8(OP_CMP_CND_SGPR) }
            else if (KIND == CMP_VCC) { CHAIN8(OP_CMP_VCC) }
            else if (KIND == CMP_SGPR) { CHAIN8(OP_CMP_SGPR) }
            else if (KIND == CMP_CND2_VCC) { CHAIN8(OP_CMP_CND2_VCC) }
            else if (KIND == CMP_CND2_E64) { CHAIN8(OP_CMP_CND2_E64) }
            else if (KIND == PK_ADD_F32) { CHAIN8(OP_PK_ADD) }
            else if (KIND == PK_MUL_F32) { CHAIN8(OP_PK_MUL) }
            else { CHAIN8(OP_SQRT_F32) }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += f[i] + (float)d[i] + (float)u[i] + (float)u2[i] + p[i].x + p[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND> void run(int n_cus, unsigned long long* d_cyc, float* d_sink, FILE* js, bool first) {
    const int iters = 4096;
    printf("%-14s", kind_name[KIND]);
    if (js) fprintf(js, "%s\n  \"%s\": {", first ? "" : ",", kind_name[KIND]);
    int col = 0;
    for (int wps : {1, 2, 4, 8}) {
        const int block = 256, grid = n_cus * wps;   // 256 threads = 4 waves = one wave per SIMD per resident block
        loop_kernel<KIND><<<grid, block>>>(d_cyc, d_sink, 16);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        loop_kernel<KIND><<<grid, block>>>(d_cyc, d_sink, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipEventDestroy(e0); hipEventDestroy(e1);
        const int n_waves = grid * 4;
        std::vector<unsigned long long> h(n_waves);
        hipMemcpy(h.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per_wave = sum / n_waves;                       // cycles one wave needed for iters * 32 instructions
        const double cyc_per_instr_simd = per_wave / (iters * 32.0) / wps;   // wps waves share the SIMD
        // the same rate from the wall clock (HIP events): ns per instruction per SIMD, and the implied tick rate
        const double ns_per_instr_simd = (double)ms * 1e6 / (iters * 32.0) / wps;
        printf("  %d w/SIMD: %5.2f cyc %5.3f ns (%.2f GHz)", wps, cyc_per_instr_simd, ns_per_instr_simd, cyc_per_instr_simd / ns_per_instr_simd);
        if (js) fprintf(js, "%s\"%d\": %.3f", col++ ? ", " : "", wps, cyc_per_instr_simd);
    }
    printf("   cycles per wave64 instruction per SIMD\n");
    if (js) fprintf(js, "}");
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    unsigned long long* d_cyc; float* d_sink;
    hipMalloc(&d_cyc, (size_t)p.multiProcessorCount * 8 * 4 * sizeof(unsigned long long));
    hipMalloc(&d_sink, (size_t)p.multiProcessorCount * 8 * 256 * sizeof(float));
    FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"unit\": \"shader cycles (s_memtime) per wave64 instruction per SIMD, by resident waves per SIMD\"", p.name, p.multiProcessorCount);
    run<FMA_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<PK_FMA_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<ADD_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<MAX_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<MAX3_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CNDMASK>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<ADD_U32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<ADD_F64>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<MUL_F64>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<FMA_F64>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<MAX_F64>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<RCP_F64>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<SQRT_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CND_VCC_SET>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CND_SGPR>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CND_SGPR_ALT>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<MIN_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<PK_ADD_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<PK_MUL_F32>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CND_E64_VCC>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CMP_CND_VCC>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CMP_CND_SGPR>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CMP_VCC>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CMP_SGPR>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CMP_CND2_VCC>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    run<CMP_CND2_E64>(p.multiProcessorCount, d_cyc, d_sink, js, false);
    if (js) { fprintf(js, "\n}\n"); fclose(js); }
    return 0;
}
