// Micro-benchmark: inside each of K blocks, does it matter WHERE the 12 written planes sit among the 31?
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
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    printf("written planes at slots: [19,31) | [0,12) | [9,21) | every other slot from 7 | 6 first + 6 last\n");
    for (int a = 0; a < K; a++) {
        float *b; CHECK(hipMalloc((void **)&b, 31 * n * 4)); CHECK(hipMemset(b, 0, 31 * n * 4));
        float t[5];
        for (int L = 0; L < 5; L++) {
            bool isout[31] = {false};
            if (L == 0) for (int j = 19; j < 31; j++) isout[j] = true;
            if (L == 1) for (int j = 0; j < 12; j++) isout[j] = true;
            if (L == 2) for (int j = 9; j < 21; j++) isout[j] = true;
            if (L == 3) for (int j = 7; j < 31; j += 2) isout[j] = true;
            if (L == 4) { for (int j = 0; j < 6; j++) isout[j] = true; for (int j = 25; j < 31; j++) isout[j] = true; }
            Planes p; int ri = 0, wi = 0;
            for (int j = 0; j < 31; j++) { if (isout[j]) p.out[wi++] = b + j * n; else p.in[ri++] = b + j * n; }
            t[L] = run(p, n);
        }
        printf("block %c: %.3f | %.3f | %.3f | %.3f | %.3f ms\n", 'A' + a, t[0], t[1], t[2], t[3], t[4]);
    }
    return 0;
}
