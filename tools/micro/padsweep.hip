// Micro-benchmark: the 19-in / 12-out plane pattern carved out of ONE allocation with a plane stride of
// n + pad floats -- same physical memory for every pad, only the relative placement of the planes changes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { const float *in[19]; float *out[12]; };

template <int XCD>
__global__ __launch_bounds__(256) void k(Planes p, long n)
{
    const long stride = (long)gridDim.x * 256;
    long first = (long)blockIdx.x * 256, step = stride, end = n;
    if (XCD) {
        const long per_xcd = n / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
        first = x * per_xcd + b * 256; step = (long)(gridDim.x / 8) * 256; end = (x + 1) * per_xcd;
    }
    for (long i = first + threadIdx.x; i < end; i += step) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 19; j++) a += __builtin_nontemporal_load(p.in[j] + i);
#pragma unroll
        for (int j = 0; j < 12; j++) __builtin_nontemporal_store(a + (float)j, p.out[j] + i);
    }
}

template <int XCD>
float run(const Planes &p, long n)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    dim3 grid(256 * 64);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<XCD>, grid, dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k<XCD>, grid, dim3(256), 0, 0, p, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}

int main(int argc, char **argv)
{
    const long n = 1L << 26;
    const long maxpad = 1L << 21;
    float *base;
    CHECK(hipMalloc((void **)&base, 31 * (n + maxpad) * 4));
    CHECK(hipMemset(base, 0, 31 * (n + maxpad) * 4));
    const long pads[] = {0, 64, 128, 256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 16384 + 64, 32768, 65536, 65536 + 1024,
                         131072, 262144, 524288, 524288 + 2048, 1048576, 1048576 + 4096 + 64, 12480, 99904, 0};
    for (long pad : pads) {
        Planes p;
        for (int j = 0; j < 19; j++) p.in[j] = base + j * (n + pad);
        for (int j = 0; j < 12; j++) p.out[j] = base + (19 + j) * (n + pad);
        float t0 = run<0>(p, n), t1 = run<1>(p, n);
        printf("pad %8ld floats (%9ld B): tile-interleaved %.3f ms %5.0f GB/s | per-XCD contiguous %.3f ms %5.0f GB/s\n", pad, pad * 4,
               t0, 124.0 * n / t0 / 1e6, t1, 124.0 * n / t1 / 1e6);
    }
    return 0;
}
