// How accurate is v_rcp_f64, and how fast are the candidate exact-division sequences?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void rcp_err(const double *x, double *r, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = __builtin_amdgcn_rcp(x[i]);
}

// variant 0: IEEE a / b;  1: (float)((double)a * rcp64((double)b))
template <int V>
__global__ __launch_bounds__(256) void divk(const float *a, const float *b, float *o, long n, int reps)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], acc = 0.f;
    for (int r = 0; r < reps; r++) {
        float q;
        if (V == 0) q = x / y;
        else q = (float)((double)x * __builtin_amdgcn_rcp((double)y));
        acc += q;
        x = q * 0.5f + x * 0.5f;     // dependent chain, keeps the optimiser honest
    }
    o[i] = acc;
}
template <int V>
__global__ __launch_bounds__(256) void div1(const float *a, const float *b, float *o, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (V == 0) o[i] = a[i] / b[i];
    else o[i] = (float)((double)a[i] * __builtin_amdgcn_rcp((double)b[i]));
}

int main()
{
    const long n = 1L << 24;
    std::vector<double> hx(n), hr(n);
    srand(1);
    for (long i = 0; i < n; i++) { double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX / RAND_MAX; hx[i] = ldexp(m, rand() % 60 - 30); }
    double *dx, *dr;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dr, n * 8));
    CHECK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rcp_err, dim3(n / 256), dim3(256), 0, 0, dx, dr, n);
    CHECK(hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (long i = 0; i < n; i++) { double e = fabs(hr[i] * hx[i] - 1.0); if (e > worst) worst = e; }
    printf("v_rcp_f64: max |x * rcp(x) - 1| = %.3g  (2^%.1f)\n", worst, log2(worst));

    // exactness of the fp64 route on random float pairs (all exponents) + timing
    std::vector<float> ha(n), hb(n), h0(n), h1(n);
    for (long i = 0; i < n; i++) {
        uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand(), v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        if (i & 1) { u = (u & 0x807fffffu) | ((uint32_t)(100 + rand() % 56) << 23); v = (v & 0x807fffffu) | ((uint32_t)(100 + rand() % 56) << 23); }
        memcpy(&ha[i], &u, 4); memcpy(&hb[i], &v, 4);
    }
    float *da, *db, *d0;
    CHECK(hipMalloc(&da, n * 4)); CHECK(hipMalloc(&db, n * 4)); CHECK(hipMalloc(&d0, n * 4));
    CHECK(hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(div1<0>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n);
    CHECK(hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(div1<1>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n);
    CHECK(hipMemcpy(h1.data(), d0, n * 4, hipMemcpyDeviceToHost));
    long bad = 0, badhost = 0;
    for (long i = 0; i < n; i++) {
        float ref = ha[i] / hb[i];
        bool n0 = isnan(h0[i]), n1 = isnan(h1[i]), nr = isnan(ref);
        if (!(n0 && nr) && memcmp(&h0[i], &ref, 4)) badhost++;
        if (!(n1 && nr) && memcmp(&h1[i], &ref, 4)) { if (bad < 5) printf("  a=%a b=%a  fp64 route %a  ieee %a\n", ha[i], hb[i], h1[i], ref); bad++; }
    }
    printf("x/y on %ld pairs: device IEEE vs host %ld mismatches; fp64-rcp route vs host %ld mismatches\n", n, badhost, bad);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int v = 0; v < 2; v++) {
        for (int w = 0; w < 2; w++) { if (v == 0) hipLaunchKernelGGL(divk<0>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n, 256); else hipLaunchKernelGGL(divk<1>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n, 256); }
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) { if (v == 0) hipLaunchKernelGGL(divk<0>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n, 256); else hipLaunchKernelGGL(divk<1>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n, 256); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d (%s): %.3f ms per launch, %.1f G divisions/s\n", v, v ? "fp64 rcp route" : "IEEE fp32 division", ms / 5, n * 256.0 / (ms / 5) / 1e6);
    }
    return 0;
}
