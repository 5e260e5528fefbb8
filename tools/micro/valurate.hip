// Issue cost of single VALU opcodes on gfx950: each kernel runs a long unrolled stream of ONE instruction over eight
// independent register chains (no dependency stalls), W waves per SIMD; cost = SIMD-cycles per wave-instruction.
// Round 5: every kernel stamps s_memtime / s_memrealtime around its loop (first wave of each workgroup), so the cost is
// printed in TRUE shader cycles (at the clock the stream actually held) beside the figure at the 2.4 GHz attribute clock
// that round 2 printed: the chip holds 1.7 GHz under a dense fp32-fma stream and 2.3 GHz under a transcendental one.
// build: hipcc --offload-arch=gfx950 -O3 -o valurate valurate.hip ; run: ./valurate
// under `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_CYCLES -- ./valurate 8` (W = 8 only) the
// per-kernel counters say what one instruction of each class does to the two counters the stall analysis rests on.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 2048, kChains = 8;

#define BODY(ASM)                                                                                        \
    float a0 = p[0] + t, a1 = p[1] + t, a2 = p[2] + t, a3 = p[3] + t, a4 = p[4] + t, a5 = p[5] + t,  \
          a6 = p[6] + t, a7 = p[7] + t;                                                                  \
    const float k = p[8], m = p[9];                                                                      \
    for (int it = 0; it < kIters; it++) {                                                                \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                             \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                     : "v"(k), "v"(m) : "vcc");                                                          \
    }                                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;

#define STAMP_BEGIN const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP_END                                                                                       \
    { const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();  \
      if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; } }
#define KERNEL(NAME, ASM)                                                                                \
    __global__ void NAME(const float *p, float *out, unsigned long long *st) { const float t = (float)threadIdx.x * 1e-9f; STAMP_BEGIN BODY(ASM) STAMP_END }

#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define A_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define A_MULU(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define A_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", 1\n"
#define A_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define A_DSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define A_DFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define A_DFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_RDLANE(i) "v_readlane_b32 s4, %" #i ", 3\n"
#define A_NOP(i) "s_nop 0\n"

KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_max, A_MAX) KERNEL(k_med3, A_MED3)
KERNEL(k_cnd, A_CND) KERNEL(k_cmp, A_CMP) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_and, A_AND) KERNEL(k_addu, A_ADDU)
KERNEL(k_mov, A_MOV) KERNEL(k_lshl, A_LSHL) KERNEL(k_bfe, A_BFE) KERNEL(k_mulu, A_MULU) KERNEL(k_cvt, A_CVT)
KERNEL(k_rcp, A_RCP) KERNEL(k_sqrt, A_SQRT) KERNEL(k_rsq, A_RSQ) KERNEL(k_exp, A_EXP) KERNEL(k_ldexp, A_LDEXP)
KERNEL(k_fract, A_FRACT) KERNEL(k_dscale, A_DSCALE) KERNEL(k_dfmas, A_DFMAS) KERNEL(k_dfix, A_DFIX)

// fp64: register pairs
#define BODY64(ASM)                                                                                      \
    double a0 = p[0] + t, a1 = p[1] + t, a2 = p[2] + t, a3 = p[3] + t, a4 = p[4] + t, a5 = p[5] + t,   \
           a6 = p[6] + t, a7 = p[7] + t;                                                                 \
    const double k = p[8], m = p[9];                                                                     \
    for (int it = 0; it < kIters; it++) {                                                                \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                             \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                     : "v"(k), "v"(m));                                                                  \
    }                                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
#define KERNEL64(NAME, ASM)                                                                              \
    __global__ void NAME(const float *p, float *out, unsigned long long *st) { const double t = (double)threadIdx.x * 1e-9; STAMP_BEGIN BODY64(ASM) STAMP_END }
#define D_FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define D_MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define D_ADD(i) "v_add_f64 %" #i ", %" #i ", %8\n"
KERNEL64(k_fma64, D_FMA) KERNEL64(k_mul64, D_MUL) KERNEL64(k_add64, D_ADD) KERNEL64(k_pkfma, A_PKFMA)

typedef void (*kern_t)(const float *, float *, unsigned long long *);
struct Case { const char *name; kern_t k; int per_chain; };

#include <algorithm>
#include <vector>
int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int khz = 0;
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const int only_w = argc > 1 ? atoi(argv[1]) : 0;          // ./valurate 8: W = 8 only (PMC runs)
    float *p, *out;
    unsigned long long *st;
    float hp[10] = {1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f, 1.8f, 0.999f, 0.001f};
    CHECK(hipMalloc(&p, sizeof(hp)));
    CHECK(hipMemcpy(p, hp, sizeof(hp), hipMemcpyHostToDevice));
    const int block = 256;                         // 4 waves: one per SIMD
    Case cases[] = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_max_f32", k_max, 1},
        {"v_med3_f32", k_med3, 1}, {"v_cndmask_b32", k_cnd, 1}, {"v_cmp_lt_f32", k_cmp, 1},
        {"v_cmp + v_cndmask", k_cmpcnd, 2}, {"v_and_b32", k_and, 1}, {"v_add_u32", k_addu, 1}, {"v_mov_b32", k_mov, 1},
        {"v_lshlrev_b32", k_lshl, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_mul_lo_u32", k_mulu, 1}, {"v_cvt_f32_u32", k_cvt, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_exp_f32", k_exp, 1},
        {"v_ldexp_f32", k_ldexp, 1}, {"v_fract_f32", k_fract, 1}, {"v_div_scale_f32", k_dscale, 1},
        {"v_div_fmas_f32", k_dfmas, 1}, {"v_div_fixup_f32", k_dfix, 1}, {"v_fma_f64", k_fma64, 1}, {"v_mul_f64", k_mul64, 1},
        {"v_add_f64", k_add64, 1}, {"v_pk_fma_f32", k_pkfma, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%d CUs, clock attribute %d MHz; per wave64 instruction per SIMD at W waves per SIMD: cycles at the attribute clock | "
           "clock held (GHz, in-kernel stamps) | TRUE cycles\n", cus, khz / 1000);
    printf("%-20s %26s %26s %26s\n", "instruction", "W=1", "W=4", "W=8");
    for (const Case &c : cases) {
        printf("%-20s", c.name);
        for (int W : {1, 4, 8}) {
            if (only_w && W != only_w) continue;
            const int blocks = cus * W;
            CHECK(hipMalloc(&out, (size_t)blocks * block * 4));
            CHECK(hipMalloc(&st, (size_t)blocks * 16));
            for (int r = 0; r < 20; r++) c.k<<<blocks, block>>>(p, out, st);       // ~0.1 s of the stream before it is timed
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) c.k<<<blocks, block>>>(p, out, st);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            std::vector<unsigned long long> h((size_t)blocks * 2);
            CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> ghz;
            for (int b = 0; b < blocks; b++) if (h[2 * b + 1]) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
            std::sort(ghz.begin(), ghz.end());
            const double clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
            const double instr_per_simd = (double)W * kIters * kChains * c.per_chain;
            const double cycles = ms * 1e-3 * (double)khz * 1e3;
            printf("   %6.2f | %5.3f GHz | %5.2f", cycles / instr_per_simd, clock, cycles / instr_per_simd * clock * 1e6 / khz);
            CHECK(hipFree(out));
            CHECK(hipFree(st));
        }
        printf("\n");
    }
    return 0;
}
