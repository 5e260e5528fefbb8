// Micro-benchmark: cache policy of the plane accesses (non-temporal or default) for the 19-in / 12-out pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { const float *in[19]; float *out[12]; };
template <int LNT, int SNT, int R, int W>
__global__ __launch_bounds__(256) void k(Planes p, long n)
{
    const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
    const long step = (long)(gridDim.x / 8) * 256, end = (x + 1) * per_xcd;
    for (long i = x * per_xcd + b * 256 + threadIdx.x; i < end; i += step) {
        float a = 1.f;
#pragma unroll
        for (int j = 0; j < R; j++) a += LNT ? __builtin_nontemporal_load(p.in[j] + i) : p.in[j][i];
#pragma unroll
        for (int j = 0; j < W; j++) { if (SNT) __builtin_nontemporal_store(a + (float)j, p.out[j] + i); else p.out[j][i] = a + (float)j; }
    }
}
template <int LNT, int SNT, int R, int W>
float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<LNT, SNT, R, W>), dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k<LNT, SNT, R, W>), dim3(256 * 64), dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}
int main()
{
    const long n = 1L << 26;
    for (int a = 0; a < 3; a++) {
        float *base; CHECK(hipMalloc((void **)&base, 31 * n * 4)); CHECK(hipMemset(base, 0, 31 * n * 4));
        Planes p;
        for (int j = 0; j < 19; j++) p.in[j] = base + j * n;
        for (int j = 0; j < 12; j++) p.out[j] = base + (19 + j) * n;
        printf("block %c  19R+12W ms: nt/nt %.3f  nt-load/plain-store %.3f  plain-load/nt-store %.3f  plain/plain %.3f | 12W GB/s: nt %4.0f plain %4.0f\n",
               'A' + a, run<1, 1, 19, 12>(p, n), run<1, 0, 19, 12>(p, n), run<0, 1, 19, 12>(p, n), run<0, 0, 19, 12>(p, n),
               48.0 * n / run<1, 1, 0, 12>(p, n) / 1e6, 48.0 * n / run<1, 0, 0, 12>(p, n) / 1e6);
    }
    return 0;
}
