// Micro-benchmark: inside ONE block, does the distance between the group of 19 read planes and the group of 12
// written planes decide the speed of the pattern?  Also: write-only and read-only rates of the same block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { const float *in[19]; float *out[12]; };

template <int R, int W>
__global__ __launch_bounds__(256) void k(Planes p, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) {
        float a = 1.f;
#pragma unroll
        for (int j = 0; j < R; j++) a += __builtin_nontemporal_load(p.in[j] + i);
#pragma unroll
        for (int j = 0; j < W; j++) __builtin_nontemporal_store(a + (float)j, p.out[j] + i);
    }
}
template <int R, int W>
float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<R, W>), dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k<R, W>), dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}
int main(int argc, char **argv)
{
    const long n = 1L << 26;
    const int K = argc > 1 ? atoi(argv[1]) : 4;
    const long slack = 3L << 26;     // three planes of slack for the gap sweep
    for (int a = 0; a < K; a++) {
        float *base; CHECK(hipMalloc((void **)&base, (31 * n + slack) * 4)); CHECK(hipMemset(base, 0, (31 * n + slack) * 4));
        Planes p;
        for (int j = 0; j < 19; j++) p.in[j] = base + j * n;
        for (int j = 0; j < 12; j++) p.out[j] = base + (19 + j) * n;
        printf("block %c: 19R+12W %.3f ms %4.0f GB/s | 19R %4.0f GB/s | 12W %4.0f GB/s | 12R+12W(copy) %4.0f GB/s\n", 'A' + a,
               run<19, 12>(p, n), 124.0 * n / run<19, 12>(p, n) / 1e6, 76.0 * n / run<19, 0>(p, n) / 1e6,
               48.0 * n / run<0, 12>(p, n) / 1e6, 96.0 * n / run<12, 12>(p, n) / 1e6);
        const long gaps[] = {0, 64, 1024, 16384, 1L << 18, 1L << 19, 1L << 20, 1L << 22, 1L << 24, 1L << 25, (1L << 26) + (1L << 25), 1L << 27};
        printf("   gap between the read group and the write group (floats -> ms):");
        for (long g : gaps) {
            for (int j = 0; j < 12; j++) p.out[j] = base + (19 + j) * n + g;
            printf("  %ld:%.3f", g, run<19, 12>(p, n));
        }
        printf("\n   interleaved R,W,R,W planes:");
        {
            int ri = 0, wi = 0;
            for (int j = 0; j < 31; j++) { if ((j % 5 == 2 || j % 5 == 4) && wi < 12) p.out[wi++] = base + j * n; else if (ri < 19) p.in[ri++] = base + j * n; else p.out[wi++] = base + j * n; }
            printf(" %.3f ms\n", run<19, 12>(p, n));
        }
    }
    // one 96-plane block (24 GB): write-only rate of every window of 12 consecutive planes, and of single planes
    {
        const int P = 96;
        float *base; CHECK(hipMalloc((void **)&base, (long)P * n * 4)); CHECK(hipMemset(base, 0, (long)P * n * 4));
        Planes p;
        for (int j = 0; j < 19; j++) p.in[j] = base;
        printf("write-only GB/s of 12-plane windows starting at plane 0, 4, 8, ...:\n  ");
        for (int w = 0; w + 12 <= P; w += 4) {
            for (int j = 0; j < 12; j++) p.out[j] = base + (long)(w + j) * n;
            printf(" %4.0f", 48.0 * n / run<0, 12>(p, n) / 1e6);
        }
        printf("\nwrite-only GB/s of single planes 0..95:\n  ");
        for (int w = 0; w < P; w++) {
            p.out[0] = base + (long)w * n;
            printf(" %4.0f", 4.0 * n / run<0, 1>(p, n) / 1e6);
            if (w % 24 == 23) printf("\n  ");
        }
        printf("\n");
    }
    return 0;
}
