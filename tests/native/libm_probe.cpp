// Which build of glibc's sinf / cosf / expf / powf does THIS host run?  glibc >= 2.28 compiles them twice on x86-64 and an
// ifunc picks the -mfma build on CPUs with AVX2 + FMA; the device libm (rlshaders_amd/csrc/rls_libm.hpp) follows that
// build by default (RLM_GLIBC_FMA = 1).  This file is compiled twice, with RLM_GLIBC_FMA = 0 and = 1, into two shared
// objects; tests/cases.py (host_libm_flavour) runs both ports and the host libm over the same random arguments and
// looks at the arguments on which the two ports differ: the libm must side with exactly one of them.  Test
// infrastructure (decides whether the parity gates can demand bit equality); built on demand with g++.
#include <math.h>
#include <stdint.h>

#include "rls_libm.hpp"

#ifndef PROBE_NAME
#error "compile with -DPROBE_NAME=libm_probe_fma or libm_probe_sse2"
#endif

// fn: 0 sinf, 1 cosf, 2 expf, 3 powf(x, 5).  port[i] = this flavour's result, libm[i] = the host libm's.
extern "C" void PROBE_NAME(int fn, int64_t n, const float *x, float *port, float *libm)
{
    static const rlm::Tables tab = RLM_TABLES_INIT;
    for (int64_t i = 0; i < n; i++) {
        const float a = x[i];
        float p, l, unused;
        switch (fn) {
        case 0: rlm::sincos32_v<true>(a, &p, &unused); l = sinf(a); break;
        case 1: rlm::sincos32_v<true>(a, &unused, &p); l = cosf(a); break;
        case 2: p = rlm::exp32(a, tab); l = expf(a); break;
        default: p = rlm::pow32(a, 5.0f, tab); l = powf(a, 5.0f); break;
        }
        port[i] = p;
        libm[i] = l;
    }
}
