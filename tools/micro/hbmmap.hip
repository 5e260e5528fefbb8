// Micro-benchmark: a map of one large allocation -- the rate at which 12 consecutive 256 MiB planes are written
// together, window by window across ~190 GB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { float *out[12]; };
__global__ __launch_bounds__(256) void k(Planes p, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) {
#pragma unroll
        for (int j = 0; j < 12; j++) __builtin_nontemporal_store((float)j, p.out[j] + i);
    }
}
float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 6; r++) hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / 6;
}
int main(int argc, char **argv)
{
    const long n = 1L << 26;                 // floats per plane (256 MiB)
    const int P = argc > 1 ? atoi(argv[1]) : 760;   // planes in the block: 760 x 256 MiB = 190 GiB
    float *base;
    CHECK(hipMalloc((void **)&base, (long)P * n * 4));
    printf("block of %d planes (%.0f GiB) at %p; 12-plane write windows every 4 planes (1 GiB), GB/s:\n", P, P / 4.0, (void *)base);
    int col = 0;
    for (int w = 0; w + 12 <= P; w += 4) {
        Planes p;
        for (int j = 0; j < 12; j++) p.out[j] = base + (long)(w + j) * n;
        printf(" %4.0f", 48.0 * n / run(p, n) / 1e6);
        if (++col % 16 == 0) printf("   <- GiB %d\n", w / 4 + 1);
    }
    printf("\n");
    return 0;
}
