// Micro-benchmark: in one 190 GiB block, (a) 12 planes written together, half of them G GiB away from the other
// half; (b) the full 19R+12W pattern with the written planes D GiB away from the read planes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { const float *in[19]; float *out[12]; };
template <int R>
__global__ __launch_bounds__(256) void k(Planes p, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) {
        float a = 1.f;
#pragma unroll
        for (int j = 0; j < R; j++) a += __builtin_nontemporal_load(p.in[j] + i);
#pragma unroll
        for (int j = 0; j < 12; j++) __builtin_nontemporal_store(a + (float)j, p.out[j] + i);
    }
}
template <int R>
float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<R>, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 6; r++) hipLaunchKernelGGL(k<R>, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / 6;
}
int main()
{
    const long n = 1L << 26;
    const int P = 760;
    float *base;
    CHECK(hipMalloc((void **)&base, (long)P * n * 4));
    CHECK(hipMemset(base, 0, (long)P * n * 4));
    auto plane = [&](double gib) { return base + (long)(gib * 4.0) * n; };      // plane index = 4 per GiB
    Planes p;
    for (int j = 0; j < 19; j++) p.in[j] = plane(8.0 + 0.25 * j);
    printf("(a) 12 planes written together: 6 at 8 GiB.., 6 at (8 + G) GiB..   G -> GB/s\n  ");
    for (double G : {1.5, 4.0, 8.0, 16.0, 24.0, 32.0, 40.0, 48.0, 52.0, 54.0, 56.0, 60.0, 64.0, 72.0, 96.0, 128.0, 160.0}) {
        for (int j = 0; j < 6; j++) { p.out[j] = plane(8.0 + 0.25 * j); p.out[6 + j] = plane(8.0 + G + 0.25 * j); }
        printf(" %.0f:%4.0f", G, 48.0 * n / run<0>(p, n) / 1e6);
    }
    printf("\n(b) 19 planes read at 8 GiB.., 12 written at (8 + D) GiB..   D -> ms of the full pattern\n  ");
    for (double D : {4.75, 8.0, 16.0, 32.0, 48.0, 52.0, 53.0, 54.0, 55.0, 56.0, 57.0, 60.0, 64.0, 96.0, 116.0, 117.0, 118.0, 120.0, 128.0}) {
        for (int j = 0; j < 12; j++) p.out[j] = plane(8.0 + D + 0.25 * j);
        printf(" %.2f:%.3f", D, run<19>(p, n));
    }
    printf("\n(c) 19R+12W with the 12 written planes split 6 / 6 across the fast boundary found in (a), reads at 8 GiB: ");
    for (int j = 0; j < 6; j++) { p.out[j] = plane(8.0 + 0.25 * j + 5.0); p.out[6 + j] = plane(8.0 + 64.0 + 0.25 * j); }
    printf("%.3f ms\n", run<19>(p, n));
    return 0;
}
