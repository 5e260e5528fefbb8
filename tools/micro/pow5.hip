// How far is the host libm's powf(x, 5) (rls_libm.hpp, pow32: glibc's fp64 log2 / exp2 tables and polynomials, 0 differences
// from glibc on all 2^32 arguments) from the correctly rounded x^5?  All x in [0, 1] (what the closures pass: a clamped
// 1 - cos).  Prints how many arguments differ, and for those how close the exact x^5 (x^5 in fp64: three products, ~1.5 ulp
// of fp64) lies to a rounding boundary of fp32 -- the distance a cheap "is this argument safe" test would have to cover.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../rlshaders_amd/csrc/rls_libm.hpp"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static __device__ __constant__ rlm::Tables c_tab = RLM_TABLES_INIT;

struct Result { unsigned long long differ, risky[8]; double maxdist_differ; unsigned int first; };

__global__ void sweep(uint32_t lo, uint32_t count, Result *res)
{
    __shared__ rlm::Tables tab;
    for (unsigned t = threadIdx.x; t < sizeof(rlm::Tables) / 8; t += blockDim.x)
        reinterpret_cast<uint64_t *>(&tab)[t] = reinterpret_cast<const uint64_t *>(&c_tab)[t];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t bits = lo + (uint32_t)i;
    const float x = __uint_as_float(bits);
    const float a = rlm::pow32(x, 5.0f, tab);
    const double d = (double)x;
    const double d2 = d * d;
    const double d5 = (d2 * d2) * d;
    const float b = (float)d5;
    // distance of d5 from the nearest fp32 rounding boundary, in fp32 ulps of the result: the 29 bits fp32 drops
    const uint64_t u = (uint64_t)__double_as_longlong(d5);
    const uint64_t low = u & ((1ull << 29) - 1);                        // 0 .. 2^29 - 1; boundary at 2^28
    const double dist = fabs((double)low - (double)(1ull << 28)) / (double)(1ull << 29);   // 0 = on the boundary, 0.5 = on a float
    const bool normal = d5 >= 1.1754943508222875e-38;                    // below that fp32 rounds on a coarser grid
    if (normal) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (dist < ldexp(1.0, -(6 + 2 * k))) atomicAdd(&res->risky[k], 1ull);      // 2^-6, 2^-8, ... 2^-20 ulp
    }
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) {
        atomicAdd(&res->differ, 1ull);
        if (normal) {
            // atomic max on a double through its bits (non-negative)
            unsigned long long *p = reinterpret_cast<unsigned long long *>(&res->maxdist_differ);
            unsigned long long old = *p, v = (unsigned long long)__double_as_longlong(dist);
            while (v > old) { unsigned long long prev = atomicCAS(p, old, v); if (prev == old) break; old = prev; }
        } else {
            atomicMax(&res->first, bits);
        }
    }
}

int main()
{
    Result *res;
    CHECK(hipMalloc(&res, sizeof(Result)));
    CHECK(hipMemset(res, 0, sizeof(Result)));
    const uint32_t hi = 0x3f800000u;                                     // 1.0f
    const uint32_t chunk = 1u << 28;
    for (uint64_t lo = 0; lo <= hi; lo += chunk) {
        const uint32_t cnt = (uint32_t)((hi + 1ull - lo) < chunk ? (hi + 1ull - lo) : chunk);
        sweep<<<(cnt + 255) / 256, 256>>>((uint32_t)lo, cnt, res);
        CHECK(hipDeviceSynchronize());
    }
    Result h;
    CHECK(hipMemcpy(&h, res, sizeof(h), hipMemcpyDeviceToHost));
    printf("x in [0, 1]: %llu arguments\n", (unsigned long long)hi + 1ull);
    printf("powf(x, 5) of the host libm != (float)(x^5 in fp64): %llu arguments\n", h.differ);
    printf("  largest distance of x^5 from an fp32 rounding boundary among them (normal results): %.3e ulp = 2^%.2f\n",
           h.maxdist_differ, log2(h.maxdist_differ));
    printf("  largest subnormal-result argument that differs: 0x%08x\n", h.first);
    for (int k = 0; k < 8; k++)
        printf("arguments whose x^5 lies within 2^-%d ulp of a boundary: %llu (%.4f %%)\n", 6 + 2 * k, h.risky[k],
               100.0 * (double)h.risky[k] / ((double)hi + 1.0));
    return 0;
}
