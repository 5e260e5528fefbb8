// Micro-benchmark: what does the 19-in / 12-out plane pattern of the reflect+refract kernel reach with
// no arithmetic, at 1, 2 and 4 floats per lane per plane?  hipcc --offload-arch=gfx950 -O3 streams31.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Planes { const float *in[19]; float *out[12]; };
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename V> struct W { static constexpr int n = sizeof(V) / 4; };

// XCD = 1: workgroups are dealt round-robin to the 8 XCDs; give each XCD one contiguous eighth of every plane
// instead of every eighth tile
template <typename V, int WORK, int XCD = 0>
__global__ __launch_bounds__(256) void k(Planes p, long n_vec)
{
    const long stride = (long)gridDim.x * 256;
    long first = (long)blockIdx.x * 256;
    long step = stride, end = n_vec;
    if (XCD) {
        const long per_xcd = n_vec / 8, x = blockIdx.x % 8, b = blockIdx.x / 8;
        first = x * per_xcd + b * 256;
        step = (long)(gridDim.x / 8) * 256;
        end = (x + 1) * per_xcd;
    }
    for (long i = first + threadIdx.x; i < end; i += step) {
        V v[19];
#pragma unroll
        for (int j = 0; j < 19; j++) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V *>(p.in[j]) + i);
        float *f = reinterpret_cast<float *>(v);
        float acc[W<V>::n];
#pragma unroll
        for (int c = 0; c < W<V>::n; c++) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 19; j++) a += f[j * W<V>::n + c];
            // WORK dependent fma per component: a stand-in for arithmetic between the loads and the stores
#pragma unroll 8
            for (int w = 0; w < WORK; w++) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
            acc[c] = a;
        }
#pragma unroll
        for (int j = 0; j < 12; j++) {
            V o;
            float *of = reinterpret_cast<float *>(&o);
#pragma unroll
            for (int c = 0; c < W<V>::n; c++) of[c] = acc[c] + (float)j;
            __builtin_nontemporal_store(o, reinterpret_cast<V *>(p.out[j]) + i);
        }
    }
}

template <typename V, int WORK, int XCD = 0>
void run(const Planes &p, long n, const char *name, int blocks_per_cu)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const long nv = n / W<V>::n;
    long want = (nv + 255) / 256, cap = 256L * blocks_per_cu;
    dim3 grid((unsigned)(want < cap ? want : cap));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<V, WORK, XCD>), grid, dim3(256), 0, 0, p, nv);
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k<V, WORK, XCD>), grid, dim3(256), 0, 0, p, nv);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-10s work %4d  blocks/CU %3d: %.3f ms  %.0f GB/s\n", name, WORK, blocks_per_cu, ms, 31.0 * 4 * n / ms / 1e6);
}

int main()
{
    const long n = 1L << 26;
    Planes p;
    for (int j = 0; j < 19; j++) { void *q; CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); p.in[j] = (const float *)q; }
    for (int j = 0; j < 12; j++) { void *q; CHECK(hipMalloc(&q, n * 4)); p.out[j] = (float *)q; }
    run<float, 0>(p, n, "dword", 64);
    run<f2, 0>(p, n, "dwordx2", 64);
    run<f4, 0>(p, n, "dwordx4", 64);
    run<float, 0>(p, n, "dword", 16);
    run<f2, 0>(p, n, "dwordx2", 16);
    run<f4, 0>(p, n, "dwordx4", 16);
    run<float, 400>(p, n, "dword", 64);
    run<f2, 400>(p, n, "dwordx2", 64);
    run<f4, 400>(p, n, "dwordx4", 64);
    run<float, 0, 1>(p, n, "dword/xcd", 64);
    run<f2, 0, 1>(p, n, "x2/xcd", 64);
    run<f4, 0, 1>(p, n, "x4/xcd", 64);
    run<float, 0, 1>(p, n, "dword/xcd", 16);
    run<float, 400, 1>(p, n, "dword/xcd", 64);
    run<float, 0>(p, n, "dword", 64);
    run<float, 800>(p, n, "dword", 64);
    run<f2, 800>(p, n, "dwordx2", 64);
    return 0;
}
