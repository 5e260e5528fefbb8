// Host check of rlshaders_amd/csrc/rls_libm.hpp: every routine against the host libm, bit for bit.
// usage: libm_faithful <count>   -> prints one line per function: name mismatches total max_ulp
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rls_libm.hpp"

static uint32_t state = 0x12345u;
static inline uint32_t rnd() { state ^= state << 13; state ^= state >> 17; state ^= state << 5; return state; }
static inline float u01() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); }
static inline int64_t ulpdiff(float a, float b)
{
    if (a == b) return 0;
    if (isnan(a) && isnan(b)) return 0;
    int32_t ia = (int32_t)rlm::f2u(a), ib = (int32_t)rlm::f2u(b);
    if (ia < 0) ia = (int32_t)0x80000000 - ia;
    if (ib < 0) ib = (int32_t)0x80000000 - ib;
    int64_t d = (int64_t)ia - ib;
    return d < 0 ? -d : d;
}

struct Acc { long bad = 0, total = 0; int64_t maxulp = 0; void add(float got, float ref) { int64_t d = ulpdiff(got, ref); total++; if (d) { bad++; if (d > maxulp) maxulp = d; } } };

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 4000000;
    static const rlm::Tables tab = RLM_TABLES_INIT;
    Acc a_exp, a_log, a_pow;
    Acc a_atan, a_atan2, a_acos, a_tan, a_sin, a_cos, v_atan, v_atan2, v_acos, v_tan, v_sin, v_cos;
    for (long i = 0; i < n; i++) {
        // expf / logf / powf
        float ex = 193.0f * u01() - 104.0f;
        a_exp.add(rlm::exp32(ex, tab), expf(ex));
        float ex2 = -30.0f * u01();
        a_exp.add(rlm::exp32(ex2, tab), expf(ex2));
        float ex3 = exp2f(-20.0f * u01()) * (rnd() & 1 ? 1.0f : -1.0f);
        a_exp.add(rlm::exp32(ex3, tab), expf(ex3));
        float lg = exp2f(60.0f * u01() - 30.0f);
        a_log.add(rlm::log32(lg, tab), logf(lg));
        float lg2 = 1.0f + (u01() - 0.5f) * exp2f(-12.0f * u01());
        a_log.add(rlm::log32(lg2, tab), logf(lg2));
        float lg3 = u01();
        a_log.add(rlm::log32(lg3, tab), logf(lg3));
        float pb = u01(), pe = u01();
        a_pow.add(rlm::pow32(pb, 5.0f, tab), powf(pb, 5.0f)); a_pow.add(rlm::pow5_32(pb, tab), powf(pb, 5.0f));
        a_pow.add(rlm::pow32(pb, pe, tab), powf(pb, pe));
        float pb2 = 100.0f * u01(), pe2 = 20.0f * u01() - 10.0f;
        a_pow.add(rlm::pow32(pb2, pe2, tab), powf(pb2, pe2));
        // atanf over many magnitudes
        float m = exp2f(40.0f * u01() - 30.0f) * (rnd() & 1 ? 1.0f : -1.0f);
        a_atan.add(rlm::atan32(m), atanf(m)); v_atan.add(rlm::atan32_v(m), atanf(m)); v_atan.add(rlm::atan32_t(m, tab), atanf(m));
        float t = 6.0f * u01() - 3.0f;
        a_atan.add(rlm::atan32(t), atanf(t)); v_atan.add(rlm::atan32_v(t), atanf(t));
        // atan2f: components of unit-ish vectors, plus scaled pairs
        float y = 2.0f * u01() - 1.0f, x = 2.0f * u01() - 1.0f;
        a_atan2.add(rlm::atan2_32(y, x), atan2f(y, x)); v_atan2.add(rlm::atan2_32_v(y, x), atan2f(y, x)); v_atan2.add(rlm::atan2_32_t(y, x, tab), atan2f(y, x)); v_atan2.add(rlm::atan2_32_q(y, x, tab), atan2f(y, x));
        float sc = exp2f(20.0f * u01() - 18.0f);
        a_atan2.add(rlm::atan2_32(y * sc, x), atan2f(y * sc, x)); v_atan2.add(rlm::atan2_32_v(y * sc, x), atan2f(y * sc, x)); v_atan2.add(rlm::atan2_32_q(y * sc, x, tab), atan2f(y * sc, x));
        a_atan2.add(rlm::atan2_32(y, x * sc), atan2f(y, x * sc)); v_atan2.add(rlm::atan2_32_v(y, x * sc), atan2f(y, x * sc)); v_atan2.add(rlm::atan2_32_q(y, x * sc, tab), atan2f(y, x * sc));
        // acosf on [-1, 1], dense near 1
        float c = 2.0f * u01() - 1.0f;
        a_acos.add(rlm::acos32(c), acosf(c)); v_acos.add(rlm::acos32_v(c), acosf(c)); v_acos.add(rlm::acos32_q(c), acosf(c));
        float c1 = 1.0f - exp2f(-24.0f * u01());
        a_acos.add(rlm::acos32(c1), acosf(c1)); v_acos.add(rlm::acos32_v(c1), acosf(c1)); v_acos.add(rlm::acos32_q(c1), acosf(c1));
        // tanf on [0, 3pi/4), dense near pi/2 and 0
        float th = 2.3561f * u01();
        a_tan.add(rlm::tan32(th), tanf(th)); v_tan.add(rlm::tan32_v(th), tanf(th)); v_tan.add(rlm::tan32_q(th), tanf(th));
        float th2 = 1.5707964f - exp2f(-22.0f * u01());
        a_tan.add(rlm::tan32(th2), tanf(th2)); v_tan.add(rlm::tan32_v(th2), tanf(th2)); v_tan.add(rlm::tan32_q(th2), tanf(th2));
        float th3 = exp2f(-20.0f * u01());
        a_tan.add(rlm::tan32(th3), tanf(th3)); v_tan.add(rlm::tan32_v(th3), tanf(th3)); v_tan.add(rlm::tan32_q(th3), tanf(th3));
        // sinf / cosf on [-2pi, 2pi] and up to +-100
        float p = 12.566371f * u01() - 6.2831855f;
        float s, co;
        rlm::sincos32(p, &s, &co);
        a_sin.add(s, sinf(p)); a_cos.add(co, cosf(p));
        rlm::sincos32_v(p, &s, &co);
        v_sin.add(s, sinf(p)); v_cos.add(co, cosf(p));
        float q = 200.0f * u01() - 100.0f;
        rlm::sincos32(q, &s, &co);
        a_sin.add(s, sinf(q)); a_cos.add(co, cosf(q));
        rlm::sincos32_v(q, &s, &co);
        v_sin.add(s, sinf(q)); v_cos.add(co, cosf(q));
        float q2 = exp2f(-16.0f * u01()) * (rnd() & 1 ? 1.0f : -1.0f);
        rlm::sincos32(q2, &s, &co);
        a_sin.add(s, sinf(q2)); a_cos.add(co, cosf(q2));
        rlm::sincos32_v(q2, &s, &co);
        v_sin.add(s, sinf(q2)); v_cos.add(co, cosf(q2));
    }
    // special values
    const float sp[] = { 0.0f, -0.0f, 1.0f, -1.0f, 0.5f, -0.5f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 0.78539816f,
                         1.5707964f, 3.1415927f, 0.75f, 0.7853981f, 1e-30f, -1e-30f, 0x1p-13f, 0x1p-12f };
    for (float v : sp) {
        a_atan.add(rlm::atan32(v), atanf(v)); v_atan.add(rlm::atan32_v(v), atanf(v));
        if (v >= -1.0f && v <= 1.0f) { a_acos.add(rlm::acos32(v), acosf(v)); v_acos.add(rlm::acos32_v(v), acosf(v)); v_acos.add(rlm::acos32_q(v), acosf(v)); }
        if (v >= 0.0f && v < 2.35f) { a_tan.add(rlm::tan32(v), tanf(v)); v_tan.add(rlm::tan32_v(v), tanf(v)); v_tan.add(rlm::tan32_q(v), tanf(v)); }
        float s, co; rlm::sincos32(v, &s, &co); a_sin.add(s, sinf(v)); a_cos.add(co, cosf(v));
        rlm::sincos32_v(v, &s, &co); v_sin.add(s, sinf(v)); v_cos.add(co, cosf(v));
        for (float w : sp) { a_atan2.add(rlm::atan2_32(v, w), atan2f(v, w)); v_atan2.add(rlm::atan2_32_v(v, w), atan2f(v, w)); v_atan2.add(rlm::atan2_32_q(v, w, tab), atan2f(v, w)); }
    }
    const float spv[] = { 0.0f, -0.0f, 1.0f, 0.5f, 2.0f, 1e-40f, 1e-38f, 88.0f, -103.0f, -104.5f, 89.0f, 3.4e38f };
    for (float v : spv) {
        a_exp.add(rlm::exp32(v, tab), expf(v));
        if (v >= 0.0f) a_log.add(rlm::log32(v, tab), logf(v));
        for (float w : spv) if (v >= 0.0f) a_pow.add(rlm::pow32(v, w, tab), powf(v, w));
    }
    printf("expf %ld %ld %lld\n", a_exp.bad, a_exp.total, (long long)a_exp.maxulp);
    printf("logf %ld %ld %lld\n", a_log.bad, a_log.total, (long long)a_log.maxulp);
    printf("powf %ld %ld %lld\n", a_pow.bad, a_pow.total, (long long)a_pow.maxulp);
    printf("atanf %ld %ld %lld\n", a_atan.bad, a_atan.total, (long long)a_atan.maxulp);
    printf("atan2f %ld %ld %lld\n", a_atan2.bad, a_atan2.total, (long long)a_atan2.maxulp);
    printf("acosf %ld %ld %lld\n", a_acos.bad, a_acos.total, (long long)a_acos.maxulp);
    printf("tanf %ld %ld %lld\n", a_tan.bad, a_tan.total, (long long)a_tan.maxulp);
    printf("sinf %ld %ld %lld\n", a_sin.bad, a_sin.total, (long long)a_sin.maxulp);
    printf("cosf %ld %ld %lld\n", a_cos.bad, a_cos.total, (long long)a_cos.maxulp);
    printf("atanf_v %ld %ld %lld\n", v_atan.bad, v_atan.total, (long long)v_atan.maxulp);
    printf("atan2f_v %ld %ld %lld\n", v_atan2.bad, v_atan2.total, (long long)v_atan2.maxulp);
    printf("acosf_v %ld %ld %lld\n", v_acos.bad, v_acos.total, (long long)v_acos.maxulp);
    printf("tanf_v %ld %ld %lld\n", v_tan.bad, v_tan.total, (long long)v_tan.maxulp);
    printf("sinf_v %ld %ld %lld\n", v_sin.bad, v_sin.total, (long long)v_sin.maxulp);
    printf("cosf_v %ld %ld %lld\n", v_cos.bad, v_cos.total, (long long)v_cos.maxulp);
    return 0;
}
