// Host side of the exhaustive elementary-function sweep (tools/libm_exhaustive.py, tests/test_gpu_libm.py):
// for each argument, (a) rls_libm.hpp compiled for the host with contraction off -- the same source the
// device runs -- and (b) the host libm itself.  Test infrastructure, built on demand with g++.
#include <math.h>
#include <stdint.h>
#include <thread>
#include <vector>

#include "rls_libm.hpp"

namespace {
const rlm::Tables kTab = RLM_TABLES_INIT;

void range(int fn, int64_t lo, int64_t hi, const float *x, const float *y, float *port, float *libm)
{
    for (int64_t i = lo; i < hi; i++) {
        const float a = x[i], b = y ? y[i] : 0.0f;
        float p, l, unused;
        switch (fn) {
        case 0: p = rlm::sqrt32(a); l = sqrtf(a); break;
        case 1: p = a / b; l = a / b; break;
        case 2: p = rlm::atan2_32_t(a, b, kTab); l = atan2f(a, b); break;
        case 3: p = rlm::acos32_v(a); l = acosf(a); break;
        case 4: case 10: p = rlm::tan32_v<true>(a); l = tanf(a); break;
        case 5: case 11: rlm::sincos32_v<true>(a, &p, &unused); l = sinf(a); break;
        case 6: case 12: rlm::sincos32_v<true>(a, &unused, &p); l = cosf(a); break;
        case 7: p = rlm::exp32(a, kTab); l = expf(a); break;
        case 8: p = rlm::log32(a, kTab); l = logf(a); break;
        case 9: p = rlm::pow32(a, b, kTab); l = powf(a, b); break;
        default: p = l = 0.0f; break;
        }
        port[i] = p;
        libm[i] = l;
    }
}
} // namespace

extern "C" void libm_host_eval(int fn, int64_t n, const float *x, const float *y, float *port, float *libm, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> pool;
    const int64_t step = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        const int64_t lo = t * step, hi = lo + step < n ? lo + step : n;
        if (lo >= hi) break;
        pool.emplace_back(range, fn, lo, hi, x, y, port, libm);
    }
    for (auto &th : pool) th.join();
}
