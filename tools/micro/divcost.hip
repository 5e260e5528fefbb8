// What does an exactly rounded fp32 division cost on gfx950, and what would a guarded fast path cost?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float div_core(float a, float b)
{
    // the compiler's sequence without v_div_scale / v_div_fmas / v_div_fixup
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float t = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(t, r, q);
}
__device__ __forceinline__ bool in_range(float x)      // 2^-62 <= |x| <= 2^62
{
    return ((__float_as_uint(x) >> 23) & 0xffu) - 65u < 125u;
}
template <int V>
__device__ __forceinline__ float dv(float a, float b)
{
    if (V == 0) return a / b;
    if (V == 1) return div_core(a, b);
    if (V == 2) return __builtin_amdgcn_div_fixupf(div_core(a, b), b, a);
    if (V == 3) {
        if (__builtin_expect(in_range(a) && in_range(b), 1)) return div_core(a, b);
        return a / b;
    }
    if (V == 4) {   // branch-free guard: select between the two
        float f = div_core(a, b);
        return (in_range(a) && in_range(b)) ? f : a / b;
    }
    if (V == 6) {   // wave-uniform guard: the whole wavefront takes the core, or the whole wavefront takes IEEE
        const uint32_t da = ((__float_as_uint(a) >> 23) & 0xffu) - 80u, db = ((__float_as_uint(b) >> 23) & 0xffu) - 80u;
        const bool ok = (da > db ? da : db) <= 94u;                      // both in [2^-47, 2^47]: v_div_scale would not scale
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) return div_core(a, b);
        return a / b;
    }
    if (V == 7) {   // reciprocal: numerator 1
        const bool ok = ((__float_as_uint(b) >> 23) & 0xffu) - 80u <= 94u;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
            float r = __builtin_amdgcn_rcpf(b);
            float e = __builtin_fmaf(-b, r, 1.0f);
            r = __builtin_fmaf(e, r, r);
            float t = __builtin_fmaf(-b, r, 1.0f);
            r = __builtin_fmaf(t, r, r);
            t = __builtin_fmaf(-b, r, 1.0f);
            return __builtin_fmaf(t, r, r);
        }
        return 1.0f / b;
    }
    return a * __builtin_amdgcn_rcpf(b);
}

template <int V>
__global__ __launch_bounds__(256) void k(const float *a, const float *b, float *o, long n, int reps)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], acc = 0.f;
    for (int r = 0; r < reps; r++) {
        float q = dv<V>(x, y);
        acc += q;
        x = q * y;                   // dependent chain that stays in range (x/y*y ~ x)
    }
    o[i] = acc;
}
// constant numerator (2 / x), varying denominator: nothing hoistable.  V 0: IEEE, 1: guard on b only + core
template <int V>
__global__ __launch_bounds__(256) void kc(const float *a, float *o, long n, int reps)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = fabsf(a[i]) + 1.0f, acc = 0.f;
    for (int r = 0; r < reps; r++) {
        float q;
        if (V == 0) q = 2.0f / x;
        else {
            const bool ok = ((__float_as_uint(x) >> 23) & 0xffu) - 80u <= 94u;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) q = div_core(2.0f, x); else q = 2.0f / x;
        }
        acc += q;
        x = q * 0.37f + 1.0f;
    }
    o[i] = acc;
}
template <int V>
void timeit_c(const char *name, const float *da, float *d0, long n)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kc<V>, dim3(n / 256), dim3(256), 0, 0, da, d0, n, 256);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kc<V>, dim3(n / 256), dim3(256), 0, 0, da, d0, n, 256);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double waves = (double)n / 64 * 256;
    printf("%-34s %.3f ms  %7.1f G div/s   ~%.1f SIMD-cycles per wave-division (incl. 2 chain ops)\n", name, ms, n * 256.0 / ms / 1e6,
           ms * 1e-3 * 2.1e9 * 1024 / waves);
}

template <int V>
__global__ __launch_bounds__(256) void one(const float *a, const float *b, float *o, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = dv<V>(a[i], b[i]);
}

template <int V>
void timeit(const char *name, const float *da, const float *db, float *d0, long n)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<V>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n, 256);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<V>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n, 256);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    // 1024 SIMDs; cycles per wave-division per SIMD at the measured rate, assuming 2.1 GHz under load
    double waves = (double)n / 64 * 256;
    printf("%-34s %.3f ms  %7.1f G div/s   ~%.1f SIMD-cycles per wave-division (incl. 3 chain ops)\n", name, ms, n * 256.0 / ms / 1e6,
           ms * 1e-3 * 2.1e9 * 1024 / waves);
}

template <int V>
long check(const char *name, const std::vector<float> &ha, const std::vector<float> &hb, const float *da, const float *db, float *d0, long n)
{
    std::vector<float> h(n);
    hipLaunchKernelGGL(one<V>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n);
    CHECK(hipMemcpy(h.data(), d0, n * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long i = 0; i < n; i++) {
        float ref = ha[i] / hb[i];
        if (!(isnan(h[i]) && isnan(ref)) && memcmp(&h[i], &ref, 4)) bad++;
    }
    printf("%-34s mismatches vs host IEEE: %ld of %ld\n", name, bad, n);
    return bad;
}

int main()
{
    const long n = 1L << 24;
    std::vector<float> ha(n), hb(n);
    srand(1);
    for (long i = 0; i < n; i++) {
        uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand(), v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        if (i & 3) { u = (u & 0x807fffffu) | ((uint32_t)(100 + rand() % 56) << 23); v = (v & 0x807fffffu) | ((uint32_t)(100 + rand() % 56) << 23); }
        memcpy(&ha[i], &u, 4); memcpy(&hb[i], &v, 4);
    }
    float *da, *db, *d0;
    CHECK(hipMalloc(&da, n * 4)); CHECK(hipMalloc(&db, n * 4)); CHECK(hipMalloc(&d0, n * 4));
    CHECK(hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice));
    printf("operands: 3/4 with exponents in [2^-27, 2^28], 1/4 raw bit patterns\n");
    check<0>("IEEE a / b", ha, hb, da, db, d0, n);
    check<1>("core (no scale / fixup)", ha, hb, da, db, d0, n);
    check<2>("core + v_div_fixup", ha, hb, da, db, d0, n);
    check<3>("guarded core, branch to IEEE", ha, hb, da, db, d0, n);
    check<4>("guarded core, select", ha, hb, da, db, d0, n);
    check<6>("wave-uniform guarded core", ha, hb, da, db, d0, n);
    // timing on in-range operands only (what the closures mostly see)
    for (long i = 0; i < n; i++) {
        uint32_t u, v; memcpy(&u, &ha[i], 4); memcpy(&v, &hb[i], 4);
        u = (u & 0x807fffffu) | ((uint32_t)(120 + (u >> 23) % 16) << 23); v = (v & 0x807fffffu) | ((uint32_t)(120 + (v >> 23) % 16) << 23);
        memcpy(&ha[i], &u, 4); memcpy(&hb[i], &v, 4);
    }
    CHECK(hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice));
    timeit<0>("IEEE a / b", da, db, d0, n);
    timeit<1>("core (no scale / fixup)", da, db, d0, n);
    timeit<2>("core + v_div_fixup", da, db, d0, n);
    timeit<3>("guarded core, branch to IEEE", da, db, d0, n);
    timeit<4>("guarded core, select", da, db, d0, n);
    timeit<6>("wave-uniform guarded core", da, db, d0, n);
    timeit<5>("a * rcp(b)  (not exact)", da, db, d0, n);
    timeit_c<0>("2 / x, IEEE", da, d0, n);
    timeit_c<1>("2 / x, guard on x + core", da, d0, n);
    return 0;
}
