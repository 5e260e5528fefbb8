// a / b through a correctly rounded reciprocal y = RN(1 / b) and Markstein's correction twice (rls_libm.hpp, div32_y)
// against the compiler's IEEE division: random operands inside the window the closures guarantee
//   2^-14 <= |b| <= 2^14,   a = 0 excluded,  2^-75 <= |a| <= 2^40
// (every intermediate stays representable there: q in [2^-89, 2^54], residuals multiples of 2^(eb + eq - 46) >= 2^-149),
// plus the corners of the window.  Prints the number of pairs whose results differ.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../rlshaders_amd/csrc/rls_libm.hpp"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h; }

__global__ void sweep(uint64_t seed, uint64_t count, int mode, unsigned long long *bad, unsigned long long *firsts)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t h1 = mix((uint32_t)(seed + i) ^ 0x9e3779b9u), h2 = mix((uint32_t)((seed + i) >> 32) + h1 * 0x85ebca6bu + 1u);
    uint32_t h3 = mix(h1 ^ (h2 << 1) ^ 0xc2b2ae35u);
    // exponent fields: b in [113, 141] (2^-14 .. 2^14), a in [52, 167] (2^-75 .. 2^40); random mantissas and signs
    uint32_t eb = 113u + h1 % 29u, ea = 52u + h2 % 116u;
    uint32_t mb = h2 >> 9, ma = h3 >> 9;
    if (mode == 1) { mb = (h2 & 1) ? 0x7fffffu - (h2 >> 28) : (h2 >> 28); ma = (h3 & 1) ? 0x7fffffu - (h3 >> 28) : (h3 >> 28); }   // mantissas near the ends
    if (mode == 2) { eb = (h1 & 1) ? 113u : 141u; ea = (h2 & 1) ? 52u : 167u; }                                                       // window corners
    const float b = __uint_as_float((h1 & 0x80000000u) | (eb << 23) | mb);
    const float a = __uint_as_float((h3 & 0x80000000u) | (ea << 23) | ma);
    const float y = rlm::rcp32_w(b);
    const float q = rlm::div32_y(a, b, y);
    const float ref = a / b;
    if (__float_as_uint(q) != __float_as_uint(ref)) {
        if (atomicAdd(bad, 1ull) < 4) { firsts[0] = __float_as_uint(a); firsts[1] = __float_as_uint(b); }
    }
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 64;          // x 2^30 pairs
    unsigned long long *bad, *firsts, h = 0, f[2] = {0, 0};
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&firsts, 16));
    CHECK(hipMemset(bad, 0, 8)); CHECK(hipMemset(firsts, 0, 16));
    const uint64_t chunk = 1ull << 30;
    for (int r = 0; r < rounds; r++) {
        sweep<<<(unsigned)(chunk / 256), 256>>>(0x1234567ull + (uint64_t)r * chunk * 3ull, chunk, r % 8 == 7 ? 2 : (r % 4 == 3 ? 1 : 0), bad, firsts);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(f, firsts, 16, hipMemcpyDeviceToHost));
    printf("div32_y vs IEEE division: %d x 2^30 pairs inside the window, %llu differ", rounds, h);
    if (h) printf(" (e.g. a = 0x%08llx, b = 0x%08llx)", f[0], f[1]);
    printf("\n");
    return 0;
}
