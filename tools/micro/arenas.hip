// Micro-benchmark: is the speed of the 19-in / 12-out plane pattern a property of WHERE in HBM the planes
// live?  K arenas of 31 planes each are allocated in one process and timed in turn, several rounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { const float *in[19]; float *out[12]; };

__global__ __launch_bounds__(256) void k(Planes p, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 19; j++) a += __builtin_nontemporal_load(p.in[j] + i);
#pragma unroll
        for (int j = 0; j < 12; j++) __builtin_nontemporal_store(a + (float)j, p.out[j] + i);
    }
}

// one plane at a time, read only: does the speed belong to a place in HBM, or to the combination of planes?
__global__ __launch_bounds__(256) void read1(const float *p, float *sink, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    float a = 0.f;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) a += __builtin_nontemporal_load(p + i);
    if (a == 12345.678f) sink[0] = a;
}
// k planes read together (k = 2, 4, 8, 16): where does the interference start?
template <int K>
__global__ __launch_bounds__(256) void readk(Planes p, float *sink, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    float a = 0.f;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) {
#pragma unroll
        for (int j = 0; j < K; j++) a += __builtin_nontemporal_load(p.in[j] + i);
    }
    if (a == 12345.678f) sink[0] = a;
}
template <typename F>
float timeit(F launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch();
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}

float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}

int main(int argc, char **argv)
{
    const long n = 1L << 26;
    const int K = argc > 1 ? atoi(argv[1]) : 6;
    const int separate = argc > 2 ? atoi(argv[2]) : 0;     // 1: 31 hipMallocs per arena, 0: one block per arena
    Planes ar[16];
    for (int a = 0; a < K; a++) {
        if (separate) {
            for (int j = 0; j < 19; j++) { void *q; CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); ar[a].in[j] = (const float *)q; }
            for (int j = 0; j < 12; j++) { void *q; CHECK(hipMalloc(&q, n * 4)); ar[a].out[j] = (float *)q; }
        } else {
            float *base; CHECK(hipMalloc((void **)&base, 31 * n * 4)); CHECK(hipMemset(base, 0, 31 * n * 4));
            for (int j = 0; j < 19; j++) ar[a].in[j] = base + j * n;
            for (int j = 0; j < 12; j++) ar[a].out[j] = base + (19 + j) * n;
        }
    }
    for (int a = 0; a < K; a++) printf("arena %c base %p (mod 1 GiB: %4lu MiB, mod 2 MiB: %4lu KiB)\n", 'A' + a, (const void *)ar[a].in[0],
                                       ((unsigned long)ar[a].in[0] >> 20) & 1023, ((unsigned long)ar[a].in[0] >> 10) & 2047);
    for (int round = 0; round < 3; round++) {
        printf("round %d (%s):", round, separate ? "31 allocations per arena" : "one allocation per arena");
        for (int a = 0; a < K; a++) { float t = run(ar[a], n); printf("  %c %.3f ms %4.0f GB/s", 'A' + a, t, 124.0 * n / t / 1e6); }
        printf("\n");
    }
    // dissect the slowest and the fastest arena
    int lo = 0, hi = 0; float tlo = 0, thi = 1e9;
    for (int a = 0; a < K; a++) { float t = run(ar[a], n); if (t > tlo) { tlo = t; lo = a; } if (t < thi) { thi = t; hi = a; } }
    float *sink; CHECK(hipMalloc((void **)&sink, 4));
    for (int which = 0; which < 2; which++) {
        const int a = which ? hi : lo;
        printf("%s arena %c (%.3f ms on the full pattern)\n  single planes, GB/s:", which ? "fastest" : "slowest", 'A' + a, which ? thi : tlo);
        for (int j = 0; j < 19; j++) {
            const float *pl = ar[a].in[j];
            float t = timeit([&] { hipLaunchKernelGGL(read1, dim3(256 * 64), dim3(256), 0, 0, pl, sink, n); });
            printf(" %4.0f", 4.0 * n / t / 1e6);
        }
        printf("\n  k planes read together, GB/s:");
        float t2 = timeit([&] { hipLaunchKernelGGL(readk<2>, dim3(256 * 64), dim3(256), 0, 0, ar[a], sink, n); });
        float t4 = timeit([&] { hipLaunchKernelGGL(readk<4>, dim3(256 * 64), dim3(256), 0, 0, ar[a], sink, n); });
        float t8 = timeit([&] { hipLaunchKernelGGL(readk<8>, dim3(256 * 64), dim3(256), 0, 0, ar[a], sink, n); });
        float t16 = timeit([&] { hipLaunchKernelGGL(readk<16>, dim3(256 * 64), dim3(256), 0, 0, ar[a], sink, n); });
        float t19 = timeit([&] { hipLaunchKernelGGL(readk<19>, dim3(256 * 64), dim3(256), 0, 0, ar[a], sink, n); });
        printf("  k=2 %4.0f  k=4 %4.0f  k=8 %4.0f  k=16 %4.0f  k=19 %4.0f\n", 8.0 * n / t2 / 1e6, 16.0 * n / t4 / 1e6, 32.0 * n / t8 / 1e6,
               64.0 * n / t16 / 1e6, 76.0 * n / t19 / 1e6);
    }
    return 0;
}
