// Proof by enumeration: candidate instruction sequences for exactly rounded ONE-argument operations (1/x, sqrt(x),
// 2/(1+sqrt(1+y))) against the compiler's IEEE forms on ALL 2^32 fp32 bit patterns.  Prints, per candidate, the number of
// arguments whose result differs (NaN == NaN), and the range of |x| over which differences occur.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../rlshaders_amd/csrc/rls_libm.hpp"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float rcp_n(float x, int steps, bool fixup)
{
    float r = __builtin_amdgcn_rcpf(x);
    for (int k = 0; k < steps; k++) {
        float e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return fixup ? __builtin_amdgcn_div_fixupf(r, x, 1.0f) : r;
}
// the compiler's own sequence minus div_scale / div_fmas (three refinements, the last two on the quotient)
__device__ __forceinline__ float rcp_core(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    float t = __builtin_fmaf(-x, q, 1.0f);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-x, q, 1.0f);
    q = __builtin_fmaf(t, r, q);
    return __builtin_amdgcn_div_fixupf(q, x, 1.0f);
}
__device__ __forceinline__ float sqrt_pm(float x)          // v_sqrt + the +-1 ulp residual correction, no rescaling
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = (0.0f >= rm) ? sm : s;
    s = (0.0f < rp) ? sp : s;
    return s;
}
__device__ __forceinline__ float sqrt_newton(float x)      // v_sqrt, one residual, one correction through v_rcp
{
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float sqrt_rsq(float x)         // Markstein: rsq, g = x y, h = y/2, two coupled steps
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float g1_ref(float y) { return 2.0f / (1.0f + sqrtf(1.0f + y)); }
__device__ __forceinline__ float g1_fast(float y) { return 2.0f * rcp_n(1.0f + sqrt_pm(1.0f + y), 2, true); }

template <int V>
__device__ __forceinline__ void eval(float x, float &got, float &ref)
{
    if (V == 0) { got = rcp_n(x, 1, true); ref = 1.0f / x; }
    if (V == 1) { got = rcp_n(x, 2, true); ref = 1.0f / x; }
    if (V == 2) { got = rcp_n(x, 3, true); ref = 1.0f / x; }
    if (V == 3) { got = rcp_core(x); ref = 1.0f / x; }
    if (V == 4) { got = rcp_n(x, 2, false); ref = 1.0f / x; }
    if (V == 5) { got = sqrt_pm(x); ref = sqrtf(x); }
    if (V == 6) { got = sqrt_newton(x); ref = sqrtf(x); }
    if (V == 7) { got = sqrt_rsq(x); ref = sqrtf(x); }
    if (V == 8) { got = g1_fast(x); ref = g1_ref(x); }
    // the library's own routines (rls_libm.hpp), as the kernels call them
    if (V == 9) { got = rlm::rcp32(x); ref = 1.0f / x; }
    if (V == 10) { got = rlm::rcp32_w(rlm::sqrt32(x)); ref = 1.0f / sqrtf(x); }                       // normalize()
    if (V == 11) { got = 2.0f * rlm::rcp32_w(1.0f + rlm::sqrt32_1p(x)); ref = 2.0f / (1.0f + sqrtf(1.0f + x)); }   // Smith G1
    if (V == 13) { float a2 = x * x; got = rlm::rcp32_hi(a2 - 1.0f); ref = 1.0f / (a2 - 1.0f); }      // 1 / (A^2 - 1)
    if (V == 14) {                                                                                   // 1 / tanf(theta), theta as in vndf_view_from
        const bool ok = x == 0.0f || (x >= 0.0141f && x <= 3.1415927f) || x != x;
        float B = rlm::tan32_v<false>(ok ? x : 1.0f);
        got = rlm::rcp32_w(B); ref = 1.0f / B;
    }
    if (V == 15) { got = rlm::sqrt32_1m(x); ref = sqrtf(1.0f - x); }
    if (V == 16) { float o = x > 1e-4f ? x : 1e-4f; got = rlm::rcp32_hi(o); ref = 1.0f / o; }          // 1 / max(ior, 1e-4)
    if (V == 17) {                                                                                   // second-slope ratio, all ry
        float u = x > 0.5f ? 2.0f * (x - 0.5f) : 2.0f * (0.5f - x);
        const float num = u * (u * (u * 0.27385f - 0.73369f) + 0.46341f);
        const float den = u * (u * (u * 0.093073f + 0.309420f) - 1.0f) + 0.597999f;
        got = !(u <= 1.0f) ? num / den : rlm::div32_m(num, den);
        ref = num / den;
    }
    if (V == 18) { got = rlm::div32_const(x, 3.0f, 1.0f / 3.0f); ref = x / 3.0f; }
    if (V == 19) { got = rlm::div32_const(x, 0.3333f, 1.0f / 0.3333f); ref = x / 0.3333f; }
    if (V == 20) { const float c = 1.0f - 0.6666f; got = rlm::div32_const(x, c, 1.0f / c); ref = x / c; }
    if (V == 21) { const float c = 0.6666f - 0.3333f; got = rlm::div32_const(x, c, 1.0f / c); ref = x / c; }
    if (V == 22) { got = !(x >= 0x1p-60f && x < 1.0f) ? x / (1.0f - x) : rlm::div32_m(x, 1.0f - x); ref = x / (1.0f - x); }
    if (V == 12) { got = rlm::rcp32(x * x) - 1.0f; ref = 1.0f / (x * x) - 1.0f; }                     // tanSqr of G1
}

struct Acc { unsigned long long bad, bad_window; unsigned int lo, hi; };   // window: 2^-126 <= |x| <= 2^126

template <int V>
__global__ __launch_bounds__(256) void sweep(Acc *acc)
{
    unsigned long long bad = 0, badw = 0;
    unsigned int lo = 0xffffffffu, hi = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
        float x = __uint_as_float((uint32_t)b), got, ref;
        eval<V>(x, got, ref);
        bool same = (__float_as_uint(got) == __float_as_uint(ref)) || (isnan(got) && isnan(ref));
        if (!same) {
            bad++;
            unsigned int a = (uint32_t)b & 0x7fffffffu;
            lo = a < lo ? a : lo; hi = a > hi ? a : hi;
            if (a >= 0x00800000u && a <= 0x7e800000u) badw++;
        }
    }
    if (bad) { atomicAdd(&acc->bad, bad); atomicAdd(&acc->bad_window, badw); atomicMin(&acc->lo, lo); atomicMax(&acc->hi, hi); }
}

template <int V>
void run(const char *name)
{
    Acc *d, h = {0, 0, 0xffffffffu, 0};
    CHECK(hipMalloc(&d, sizeof(Acc)));
    CHECK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sweep<V>, dim3(256 * 64), dim3(256), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
    float flo, fhi;
    memcpy(&flo, &h.lo, 4); memcpy(&fhi, &h.hi, 4);
    if (h.bad) printf("%-44s %12llu differ (%llu of them with 2^-126 <= |x| <= 2^126), |x| in [%.9g (0x%08x), %.9g (0x%08x)]\n", name, h.bad, h.bad_window, flo, h.lo, fhi, h.hi);
    else printf("%-44s            0 differ on all 2^32 arguments\n", name);
    CHECK(hipFree(d));
}

int main()
{
    run<0>("1/x: rcp + 1 Newton step + fixup");
    run<1>("1/x: rcp + 2 Newton steps + fixup");
    run<2>("1/x: rcp + 3 Newton steps + fixup");
    run<3>("1/x: compiler core (no scale/fmas) + fixup");
    run<4>("1/x: rcp + 2 Newton steps, no fixup");
    run<5>("sqrt: v_sqrt +-1ulp correction, no rescale");
    run<6>("sqrt: v_sqrt + one Newton step via v_rcp");
    run<7>("sqrt: v_rsq Markstein");
    run<8>("2/(1+sqrt(1+y)): fast forms");
    run<9>("rlm::rcp32(x) vs 1/x");
    run<10>("rlm::rcp32_w(sqrt32(x)) vs 1/sqrtf(x)");
    run<11>("2*rcp32_w(1+sqrt32_1p(y)) vs 2/(1+sqrtf(1+y))");
    run<12>("rcp32(x*x)-1 vs 1/(x*x)-1");
    run<13>("rcp32_hi(A*A-1) vs 1/(A*A-1)");
    run<15>("sqrt32_1m(t) vs sqrtf(1-t)");
    run<18>("div32_const(x, 3) vs x / 3");
    run<19>("div32_const(x, 0.3333) vs x / 0.3333");
    run<20>("div32_const(x, 1 - 0.6666)");
    run<21>("div32_const(x, 0.6666 - 0.3333)");
    run<22>("rx / (1 - rx): div32_m under 2^-60 <= rx < 1");
    run<17>("slope ratio num(u)/den(u), u = 2|ry-1/2|: div32_m under u <= 1");
    run<16>("rcp32_hi(max(ior,1e-4)) vs 1/max(ior,1e-4)");
    run<14>("rcp32_w(tanf(theta)) vs 1/tanf(theta), theta in {0} U [.0141, pi]");
    return 0;
}
