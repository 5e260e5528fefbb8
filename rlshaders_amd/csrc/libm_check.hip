// libm_check.hip -- which host libm the EXACT kernels reproduce, and whether the CALLER's libm is that one (host code only).
//
// "Bit for bit like the CPU closures" is a statement about one libm: the reference's closures call sinf / cosf / expf /
// logf / powf / atan2f / acosf / tanf of whatever C library the renderer runs on (src/rlGgx.cpp:27-58, src/rlDisney.cpp:177,
// 399,549,576, src/rlSss.cpp:31-32,59,62,78-79,102); csrc/rls_libm.hpp restates glibc's algorithms (verified range 2.28 <=
// glibc < 2.41, read from 2.35; 2.41's correctly rounded tanf / acosf / atan2f are a different libm and fail part 2 below), and
// of the two x86-64 builds glibc ships the one RLM_GLIBC_FMA selects at compile time.  A host with another C library (a newer
// glibc, musl, MSVC's UCRT, glibc on a CPU without FMA when the library follows the FMA build) gets the alternate-libm tail
// of SURVEY.md Appendix D (0.009-0.14 % of chained outputs beyond 1e-5), not bit-identical results; these two entry points
// let a host ask instead of assume.
#include <math.h>

#include "rls_internal.hpp"

namespace {

struct FlavourArg { int fn; uint32_t x, fma, sse2; };
const FlavourArg kFlavourArgs[] = {
#include "rls_libm_flavour_args.inc"
};

// the process's libm through volatile pointers: no constant folding, no compiler builtin standing in for the call
float (*volatile p_sinf)(float) = sinf;
float (*volatile p_cosf)(float) = cosf;
float (*volatile p_expf)(float) = expf;
float (*volatile p_logf)(float) = logf;
float (*volatile p_powf)(float, float) = powf;
float (*volatile p_atan2f)(float, float) = atan2f;
float (*volatile p_acosf)(float) = acosf;
float (*volatile p_tanf)(float) = tanf;
float (*volatile p_sqrtf)(float) = sqrtf;

inline bool same(float a, float b) { return rlm::f2u(a) == rlm::f2u(b) || (a != a && b != b); }

} // namespace

extern "C" {

const char *rls_libm_flavour(void)
{
    return RLM_GLIBC_FMA ? "glibc-fma" : "glibc-sse2";
}

rls_status rls_host_libm_matches(int *mismatches)
{
    RLS_REQUIRE(mismatches != nullptr, "mismatches is NULL");
    static const rlm::Tables tab = RLM_TABLES_INIT;
    int bad = 0;
    // 1. the arguments on which glibc's two builds differ: the host must side with the build this library follows
    for (const FlavourArg &a : kFlavourArgs) {
        const float x = rlm::u2f(a.x);
        const float got = a.fn == 0 ? p_sinf(x) : a.fn == 1 ? p_cosf(x) : a.fn == 2 ? p_expf(x) : p_powf(x, 5.0f);
        if (rlm::f2u(got) != (RLM_GLIBC_FMA ? a.fma : a.sse2)) bad++;
    }
    // 2. a libm that is neither build (another C library): this library's own routines, compiled for the host, against the
    //    host's on 4096 arguments per function over the closures' ranges
    uint32_t s = 0x2545f491u;
    auto u01 = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < 4096; i++) {
        const float ang = 12.566371f * u01() - 6.2831855f;
        float sn, cs;
        rlm::sincos32(ang, &sn, &cs);
        bad += !same(sn, p_sinf(ang)) + !same(cs, p_cosf(ang));
        const float e = -40.0f * u01();
        bad += !same(rlm::exp32(e, tab), p_expf(e));
        const float l = u01() + 0x1p-24f;
        bad += !same(rlm::log32(l, tab), p_logf(l));
        const float b = u01(), y = u01();
        bad += !same(rlm::pow5_32(b, tab), p_powf(b, 5.0f)) + !same(rlm::pow32(b, y, tab), p_powf(b, y));
        const float ay = 2.0f * u01() - 1.0f, ax = 2.0f * u01() - 1.0f;
        bad += !same(rlm::atan2_32(ay, ax), p_atan2f(ay, ax));
        const float c = 2.0f * u01() - 1.0f;
        bad += !same(rlm::acos32(c), p_acosf(c));
        const float th = 1.5707f * u01();
        bad += !same(rlm::tan32(th), p_tanf(th));
        const float q = 4.0f * u01();
        bad += !same(rlm::sqrt32(q), p_sqrtf(q));
    }
    *mismatches = bad;
    return RLS_OK;
}

} // extern "C"
