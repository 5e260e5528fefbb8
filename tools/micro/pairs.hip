// Micro-benchmark: 19 planes read and 6 written in block X, the other 6 written in block Y, for every pair (X, Y) of
// K separately allocated blocks: how often does a pair beat the best single block?
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
        float a = 1.f;
#pragma unroll
        for (int j = 0; j < 19; j++) a += __builtin_nontemporal_load(p.in[j] + i);
#pragma unroll
        for (int j = 0; j < 12; j++) __builtin_nontemporal_store(a + (float)j, p.out[j] + i);
    }
}
float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / 5;
}
int main(int argc, char **argv)
{
    const long n = 1L << 26;
    const int K = argc > 1 ? atoi(argv[1]) : 10;
    float *blk[32];
    for (int a = 0; a < K; a++) { CHECK(hipMalloc((void **)&blk[a], 31 * n * 4)); CHECK(hipMemset(blk[a], 0, 31 * n * 4)); }
    float single[32]; int best = 0;
    printf("single blocks (ms):");
    for (int a = 0; a < K; a++) {
        Planes p;
        for (int j = 0; j < 19; j++) p.in[j] = blk[a] + j * n;
        for (int j = 0; j < 12; j++) p.out[j] = blk[a] + (19 + j) * n;
        single[a] = run(p, n);
        printf(" %.3f", single[a]);
        if (single[a] < single[best]) best = a;
    }
    printf("\nbest single: %c %.3f\npairs (row X = reads + 6 writes, column Y = other 6 writes), ms:\n", 'A' + best, single[best]);
    float bestpair = 1e9; int bx = 0, by = 0;
    for (int x = 0; x < K; x++) {
        printf("  %c:", 'A' + x);
        for (int y = 0; y < K; y++) {
            if (x == y) { printf("   -  "); continue; }
            Planes p;
            for (int j = 0; j < 19; j++) p.in[j] = blk[x] + j * n;
            for (int j = 0; j < 6; j++) { p.out[j] = blk[x] + (19 + j) * n; p.out[6 + j] = blk[y] + j * n; }
            float t = run(p, n);
            printf(" %.3f", t);
            if (t < bestpair) { bestpair = t; bx = x; by = y; }
        }
        printf("\n");
    }
    printf("best pair: X=%c Y=%c %.3f ms (%.0f GB/s) vs best single %.3f ms\n", 'A' + bx, 'A' + by, bestpair, 124.0 * n / bestpair / 1e6, single[best]);
    return 0;
}
