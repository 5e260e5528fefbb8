// rls_libm.hpp -- fp32 elementary functions that reproduce the HOST libm bit for bit.
//
// Why: the reference's closures call atan2f / acosf / tanf / sinf / cosf (src/rlGgx.cpp:71-91,
// src/rlDisney.cpp:474-494 of the reference) and feed the results into the ill-conditioned slope
// equations of visible-normal sampling, where a 1-ulp change of an angle moves 0.1-0.5 % of the
// outputs by more than 1e-5.  ROCm's device libm (ocml) rounds these
// functions differently from the libm the reference's CPU build links (glibc on Linux), so the
// only way to make the GPU closures agree with the CPU closures on every point -- not just on
// the well-conditioned ones -- is to compute the angles with the same algorithms:
//   * atanf / atan2f / acosf / tanf: the fdlibm single-precision algorithms (notice below) that glibc <= 2.40 ships in
//     sysdeps/ieee754/flt-32/{s_atanf,e_atan2f,e_acosf,s_tanf,k_tanf,e_rem_pio2f}.c -- pure
//     fp32 + - * / sqrt sequences, reproducible exactly with FMA contraction off;
//   * sinf / cosf: the double-precision polynomial scheme of glibc >= 2.28 (s_sincosf.h, from
//     ARM's optimized-routines): reduce by pi/2 in fp64, degree-7/8 polynomials in fp64, round
//     once to fp32.  (glibc's FMA-multiarch build may contract the fp64 steps; the final fp32
//     rounding hides that on all but ~1e-8 of arguments.)
// Restated here from the published algorithms for the argument ranges the closures produce
// (|x| <= 2*pi for sin/cos, [0, pi/2] for tan, [-1, 1] for acos); outside them the routines
// still return a correct value but fall back to simple forms.
//
// The header compiles for the host too (tests/test_libm_faithful.py builds it with g++ and
// compares every routine against the host libm over millions of arguments, bit for bit).
//
// Provenance and licences of what is restated here: THIRD_PARTY.md at the repository root.  The fdlibm routines (atan32,
// atan2_32, acos32, kernel_tan32 / tan32 and their variants below) derive from code that carries this notice:
/*
 * ====================================================
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunPro, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ====================================================
 */
// The sinf / cosf / expf / logf / powf routines restate Arm Optimized Routines (Copyright (c) 2017-2018, Arm Limited; MIT);
// glibc contributed observed behaviour only (which steps its FMA build fuses), no code.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RLM_FN __host__ __device__ __forceinline__
#else
#define RLM_FN static inline
#endif

namespace rlm {

RLM_FN uint32_t f2u(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(x);
#else
    uint32_t u; memcpy(&u, &x, 4); return u;
#endif
}
RLM_FN float u2f(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float x; memcpy(&x, &u, 4); return x;
#endif
}
RLM_FN float fabs32(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// a / b: the compiler's exactly rounded IEEE sequence.  (Its arithmetic core without v_div_scale / v_div_fmas behind an
// operand-window test was measured twice -- round 1 in isolation, round 2 in the kernels -- and is slower: the test costs
// what the three special instructions cost.  profiles/r01_divcost.txt, profiles/r02_exact1.txt.)
RLM_FN float div32(float a, float b) { return a / b; }

// ---- which build of glibc's fp64-polynomial routines to reproduce --------------------------------------------
// glibc >= 2.28 compiles sinf / cosf / sincosf / expf / logf / powf twice on x86-64: a baseline SSE2 build and
// an -mfma -mavx2 build (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c, e_expf-fma.c, ...) and an ifunc picks the FMA
// build on every CPU that has AVX2 + FMA -- every x86 server since 2013, the GPU box's EPYC host included.  That
// is the code the reference's CPU closures (and the oracle) actually execute there.  GCC contracts every
// `a * b + c` of those sources; the pattern below was read from the disassembly of the FMA entry points of
// glibc 2.35 (the resolvers of sinf / cosf / expf / logf / powf in libm.so.6 name them): each polynomial step is one
// fma, the pi/2 reduction is x - n*hpi as one fma, and expf forms both kd = InvLn2N*x + Shift and
// r = InvLn2N*x - kd as fmas of the unrounded product.  RLM_GLIBC_FMA = 1 (default) follows that build,
// 0 the uncontracted SSE2 build (a host without FMA); the two differ in the last fp32 bit on ~1e-8 of arguments.
// tanf / atanf / atan2f / acosf are fdlibm fp32 code, built once, never contracted.
#ifndef RLM_GLIBC_FMA
#define RLM_GLIBC_FMA 1
#endif
RLM_FN double mad(double a, double b, double c)
{
#if RLM_GLIBC_FMA
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}

// Correctly rounded fp32 square root.  Host: libm's sqrtf.  Device: the compiler's expansion of
// sqrtf spends 7 of its 16 instructions rescaling arguments below 2^-96 so that the residuals of
// its final +-1 ulp correction stay normal; the closures' radicands are never that small, so the
// common path here is the same v_sqrt_f32 + correction without the rescaling, and arguments
// below 2^-96 (never seen in practice) take the compiler's full sequence.  Identical results.
// GUARDED = false: for radicands of the form 1 + y.  The sum of 1 and an fp32 number is a multiple of 2^-24 (or
// exceeds 2), so it is never in (0, 2^-96) and the rescaling branch below is dead whatever y is; leaving the
// test out saves three vector instructions and a branch per call.
template <bool GUARDED = true>
RLM_FN float sqrt32(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // |x| < 2^-96, x != 0 (negative subnormals included: v_sqrt_f32 would flush them to -0 instead of NaN)
    if (GUARDED && __builtin_expect((f2u(x) & 0x7fffffffu) - 1u < 0x0f800000u - 1u, 0)) return sqrtf(x);
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u);
    const float sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    s = (0.0f >= rm) ? sm : s;
    s = (0.0f < rp) ? sp : s;
    return s;
#else
    return sqrtf(x);
#endif
}
RLM_FN float sqrt32_1p(float y) { return sqrt32<false>(1.0f + y); }   // sqrtf(1 + y)
// sqrtf(1 - t): the difference is exact for t in [1/2, 2] (a multiple of ulp(t) >= 2^-25, or 0), at least 1/2 for smaller
// t and negative beyond 2 -- never in (0, 2^-96) either, so no guard (all 2^32 arguments: tools/micro/exact1.hip)
RLM_FN float sqrt32_1m(float t) { return sqrt32<false>(1.0f - t); }

// Exactly rounded reciprocal.  v_rcp_f32 is accurate to 1 ulp; ONE Newton step e = 1 - x r, r' = r + e r (two fmas)
// then rounds to the correctly rounded 1/x for EVERY x with 2^-126 <= |x| <= 2^126 -- checked by enumeration of all
// 2^32 bit patterns on gfx950 (tools/micro/exact1.hip, profiles/r02_exact1.txt: 0 differences from the compiler's IEEE
// division inside the window; outside it the reciprocal or its residual is subnormal and the sequence is wrong).
// v_div_fixup_f32 supplies the IEEE results for x = +-0, +-inf and NaN.  4 instructions instead of the 11 of `1.0f / x`
// (2 v_div_scale, v_rcp, 5 fma/mul, v_div_fmas, v_div_fixup), and none of them slow to issue.
// rcp32_w: for arguments known to be 0, inf, NaN or inside the window (a square root, 1 + a square root, ...).
// rcp32: any argument -- subnormal or huge ones (two compares) take the IEEE sequence.
RLM_FN float rcp32_w(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
#else
    return 1.0f / x;
#endif
}
RLM_FN float rcp32(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // +-subnormal (class bits 4 and 7), or beyond 2^126 (1/x subnormal; +-inf lands here too and is slow but right)
    if (__builtin_expect(__builtin_amdgcn_classf(x, 0x90) || __builtin_fabsf(x) > 0x1p126f, 0)) return 1.0f / x;
    return rcp32_w(x);
#else
    return 1.0f / x;
#endif
}

// a / b inside the elementary functions below, where the denominator is a polynomial or sum that stays far inside
// [2^-126, 2^126] for every argument whose result is not replaced afterwards: Markstein's short division -- the exactly
// rounded reciprocal y (above), q0 = a y, the exact residual r = a - b q0, q = q0 + r y -- 6 instructions.  That it
// rounds like IEEE division is not assumed from the theorem (q0 need not be faithful) but checked where it is used:
// tools/libm_exhaustive.py runs atanf, acosf and tanf on all 2^32 arguments against the host libm.
RLM_FN float div32_m(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
#else
    return a / b;
#endif
}

// x / C for a compile-time constant C: the short division with the reciprocal rounded at compile time, 3 instructions
// for 2^-100 <= |x| <= 2^100 (the residual x - C q0 is then exact), IEEE division outside (zeros included: the
// sign of -0 / C would be lost).  Used for the constants
// tools/micro/exact1.hip has run over all 2^32 numerators: 3, 0.3333, 1 - 0.6666, 0.6666 - 0.3333.
// the same without the range test, for callers whose x is inside 2^-100 .. 2^100 by construction (and not zero)
RLM_FN float div32_const_w(float x, float c, float rc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float q0 = x * rc;
    const float r = __builtin_fmaf(-c, q0, x);
    return __builtin_fmaf(r, rc, q0);
#else
    (void)rc;
    return x / c;
#endif
}
RLM_FN float div32_const(float x, float c, float rc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(!(__builtin_fabsf(x) >= 0x1p-100f && __builtin_fabsf(x) <= 0x1p100f), 0)) return x / c;   // 0, NaN too
    const float q0 = x * rc;
    const float r = __builtin_fmaf(-c, q0, x);
    return __builtin_fmaf(r, rc, q0);
#else
    (void)rc;
    return x / c;
#endif
}

// rcp32 for arguments that are 0, NaN or at least 2^-126 in magnitude by construction: only the upper end is tested
RLM_FN float rcp32_hi(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(!(__builtin_fabsf(x) <= 0x1p126f), 0)) return 1.0f / x;
    return rcp32_w(x);
#else
    return 1.0f / x;
#endif
}

// ---- atanf: fdlibm s_atanf.c ---------------------------------------------------------------------
RLM_FN float atan32(float x)
{
    const float atanhi0 = u2f(0x3eed6338u), atanhi1 = u2f(0x3f490fdau), atanhi2 = u2f(0x3f7b985eu),
                atanhi3 = u2f(0x3fc90fdau);
    const float atanlo0 = u2f(0x31ac3769u), atanlo1 = u2f(0x33222168u), atanlo2 = u2f(0x33140fb4u),
                atanlo3 = u2f(0x33a22168u);
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u), aT3 = u2f(0xbde38e38u),
                aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u), aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu),
                aT8 = u2f(0x3d4bda59u), aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {                       // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;        // NaN
        return hx > 0 ? atanhi3 + atanlo3 : -atanhi3 - atanlo3;
    }
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix < 0x3ee00000) {                        // |x| < 0.4375
        if (ix < 0x31000000) return x;            // |x| < 2^-29
        id = -1;
    } else {
        x = fabs32(x);
        if (ix < 0x3f980000) {                    // |x| < 1.1875
            if (ix < 0x3f300000) {                // 7/16 <= |x| < 11/16
                id = 0; hi = atanhi0; lo = atanlo0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {                              // 11/16 <= |x| < 19/16
                id = 1; hi = atanhi1; lo = atanlo1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000) {                // |x| < 2.4375
                id = 2; hi = atanhi2; lo = atanlo2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {                              // 2.4375 <= |x| < 2^25
                id = 3; hi = atanhi3; lo = atanlo3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

// ---- atan2f: fdlibm e_atan2f.c ---------------------------------------------------------------------
RLM_FN float atan2_32(float y, float x)
{
    const float tiny = 1.0e-30f;
    const float pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu), pi_lo = u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;              // NaN
    if (hx == 0x3f800000) return atan32(y);                               // x == 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                    // 2*sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
        case 0:
        case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                                // |y/x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                                 // |y|/x < -2^60
    else z = atan32(fabs32(y / x));
    switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---- acosf: fdlibm e_acosf.c ----------------------------------------------------------------------
RLM_FN float acos32(float x)
{
    const float pi = u2f(0x40490fdau), pio2_hi = u2f(0x3fc90fdau), pio2_lo = u2f(0x33a22168u);
    const float pS0 = u2f(0x3e2aaaabu), pS1 = u2f(0xbea6b090u), pS2 = u2f(0x3e4e0aa8u), pS3 = u2f(0xbd241146u),
                pS4 = u2f(0x3a4f7f04u), pS5 = u2f(0x3811ef08u);
    const float qS1 = u2f(0xc019d139u), qS2 = u2f(0x4001572du), qS3 = u2f(0xbf303361u), qS4 = u2f(0x3d9dc62eu);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;    // |x| == 1
    if (ix > 0x3f800000) return (x - x) / (x - x);                        // |x| > 1: NaN
    if (ix < 0x3f000000) {                                                // |x| < 0.5
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;                   // |x| < 2^-57
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {                                                         // x < -0.5
        const float z = (1.0f + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrt32(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (1.0f - x) * 0.5f;                                    // x > 0.5
    const float s = sqrt32(z);
    const float df = u2f(f2u(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// ---- tanf: fdlibm s_tanf.c + k_tanf.c; argument reduction as glibc >= 2.27 e_rem_pio2f.c (fp64) -----
RLM_FN float kernel_tan32(float x, float y, int iy)
{
    const float pio4 = u2f(0x3f490fdau), pio4lo = u2f(0x33222168u);
    const float T0 = u2f(0x3eaaaaabu), T1 = u2f(0x3e088889u), T2 = u2f(0x3d5d0dd1u), T3 = u2f(0x3cb327a4u),
                T4 = u2f(0x3c11371fu), T5 = u2f(0x3b6b6916u), T6 = u2f(0x3abede48u), T7 = u2f(0x3a1a26c8u),
                T8 = u2f(0x398137b9u), T9 = u2f(0x38a3f445u), T10 = u2f(0x3895c07au), T11 = u2f(0xb79bae5fu),
                T12 = u2f(0x37d95384u);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {                                                // |x| < 2^-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f / fabs32(x);
            if (iy == 1) return x;
            return -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) {                                               // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        const float zz = pio4 - x;
        const float ww = pio4lo - y;
        x = zz + ww;
        y = 0.0f;
        if (fabs32(x) < 0x1p-13f) return (float)(1 - ((hx >> 30) & 2)) * (float)iy * (1.0f - 2.0f * (float)iy * x);
    }
    float z = x * x;
    float w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1/(x+r) computed accurately
    z = u2f(f2u(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = u2f(f2u(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}


// ---- reduce_large of glibc's s_sincosf.h: |x| >= 120, against 192 bits of 4/pi -------------------------
// Returns |x| reduced to [-pi/4, pi/4] and the quadrant; never reached by the closures (their angles come out
// of atan2f / acosf / 2 pi xi), so whole wavefronts skip it.
RLM_FN double reduce_large(uint32_t xi, int *np)
{
    static const uint32_t inv_pio4[24] = {
        0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
        0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
        0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u };
    const double pi63 = 0x1.921FB54442D18p-62;
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (int)(xi >> 23) & 7;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * pi63;
}

// ---- sinf / cosf: glibc >= 2.28 s_sincosf.h ----------------------------------------------------------
// Returns both values; each equals what sinf(x) / cosf(x) return separately.
RLM_FN void sincos32(float y, float *sinp, float *cosp)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23;   // 2/pi * 2^24
    const double hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
    double x = (double)y;
    int n = 0, m = 0;
    bool tiny = false;
    if (top < 0x3f4u) {                                                   // abstop12(y) < abstop12(pi/4)
        tiny = top < 0x398u;                                              // |y| < 2^-12: sin = y, cos = 1
    } else if (top < 0x42fu) {                                            // |y| < 120
        const double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = mad(-(double)n, hpi, x);
    } else if (top < 0x7f8u) {                                            // |y| >= 120, finite
        x = reduce_large(f2u(y), &n);
        m = n + (int)(f2u(y) >> 31);                                      // signs include the argument's
    } else {
        *sinp = y - y;                                                    // inf or NaN
        *cosp = y - y;
        return;
    }
    if (top < 0x42fu) m = n;
    // quadrant handling of sinf/cosf: sign[m&3] applied to x, second table (negated cosine) if m&2
    const double sgn = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
    const double cs = (m & 2) ? -1.0 : 1.0;
    const double xs = x * sgn;
    const double x2 = x * x;
    // sine polynomial in xs
    const double x3 = xs * x2;
    const double s1 = mad(x2, S3, S2);
    const double x7 = x3 * x2;
    const double sv = mad(x3, S1, xs);
    const double sres = mad(x7, s1, sv);
    // cosine polynomial (table[1] negates every coefficient)
    const double x4 = x2 * x2;
    const double c2 = mad(x2, cs * C4, cs * C3);
    const double c1 = mad(x2, cs * C1, cs * C0);
    const double x6 = x4 * x2;
    const double cv = mad(x4, cs * C2, c1);
    const double cres = mad(x6, c2, cv);
    // sinf(y): quadrant n -> sine poly if n even else cosine poly; cosf(y): uses n ^ 1
    // Note the sign conventions of s_sinf.c / s_cosf.c: both multiply x by sign[n & 3] and select
    // the second table on n & 2; the polynomial is chosen by the parity of n (sin) or n ^ 1 (cos).
    float sf, cf;
    if ((n & 1) == 0) { sf = (float)sres; cf = (float)cres; }
    else              { sf = (float)cres; cf = (float)sres; }
    if (tiny) { sf = y; cf = 1.0f; }
    *sinp = sf;
    *cosp = cf;
}

RLM_FN float tan32(float x)
{
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return kernel_tan32(x, 0.0f, 1);                // |x| <= pi/4
    if (ix < 0x42f00000) {                                                // |x| < 120
        // __ieee754_rem_pio2f of glibc >= 2.27: the fp64 reduction of s_sincosf.h, split into
        // a high and a low fp32 part
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double r = (double)x * hpi_inv;
        const int n = ((int32_t)r + 0x800000) >> 24;
        const double dx = (double)x - (double)n * hpi;
        const float y0 = (float)dx;
        const float y1 = (float)(dx - (double)y0);
        return kernel_tan32(y0, y1, 1 - ((n & 1) << 1));
    }
    if (ix >= 0x7f800000) return x - x;                                   // inf or NaN
    int n;
    double dx = reduce_large(f2u(x), &n);                                 // __ieee754_rem_pio2f, large branch
    if (hx < 0) { dx = -dx; n = -n; }
    const float y0 = (float)dx;
    const float y1 = (float)(dx - (double)y0);
    return kernel_tan32(y0, y1, 1 - ((n & 1) << 1));
}

// =================================================================================================
// expf / logf / powf: the fp64 table-driven algorithms of glibc >= 2.28 (ARM optimized-routines:
// e_expf.c, e_logf.c, e_powf.c).  The tables (rls_libm_tables.inc, generated and cross-checked by
// tools/gen_libm_tables.py) are passed in: a static object on the host, an LDS copy on the device
// (each is exactly one 256-byte LDS bank row or less, so per-lane lookups are conflict-free).
// =================================================================================================
#include "rls_libm_tables.inc"

// atan_k: the four reduction ranges of fdlibm's atanf beyond |x| < 7/16, one row each: t = (A|x| + B) / (C|x| + D),
// atanhi, atanlo (s_atanf.c).  Row k is 7/16 <= |x| < 11/16, < 19/16, < 39/16, >= 39/16.
struct Tables {
    uint64_t exp2t[32];
    double invc[16], logc[16], log2c[16];
    float atan_k[4][8];
};
#define RLM_ATAN_K { { 2.0f, -1.0f, 1.0f, 2.0f, 0x1.dac670p-2f, 0x1.586ed2p-28f, 0.0f, 0.0f }, \
                     { 1.0f, -1.0f, 1.0f, 1.0f, 0x1.921fb4p-1f, 0x1.4442d0p-25f, 0.0f, 0.0f }, \
                     { 1.0f, -1.5f, 1.5f, 1.0f, 0x1.f730bcp-1f, 0x1.281f68p-25f, 0.0f, 0.0f }, \
                     { 0.0f, -1.0f, 1.0f, 0.0f, 0x1.921fb4p+0f, 0x1.4442d0p-24f, 0.0f, 0.0f } }
#define RLM_TABLES_INIT { RLM_EXP2_TABLE, RLM_LOG_INVC, RLM_LOG_LOGC, RLM_LOG_LOG2C, RLM_ATAN_K }

RLM_FN uint64_t d2u(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(x);
#else
    uint64_t u; memcpy(&u, &x, 8); return u;
#endif
}
RLM_FN double u2d(uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double x; memcpy(&x, &u, 8); return x;
#endif
}

// s * (C0 r^3 + C1 r^2 + C2 r + 1) with s = 2^(k/32) from the table: the tail shared by expf and powf
RLM_FN float exp2_tail(double r, uint64_t ki, double c0, double c1, double c2, const Tables &t, uint64_t sign_bias)
{
    uint64_t tt = t.exp2t[ki % 32];
    tt += (ki + sign_bias) << (52 - 5);
    const double s = u2d(tt);
    const double zz = mad(c0, r, c1);
    const double r2 = r * r;
    double y = mad(c2, r, 1.0);
    y = mad(zz, r2, y);
    y = y * s;
    return (float)y;
}

// expf behind its range tests: for |x| < 88 this is all of expf; a NaN comes out as a NaN (the fma chain carries it; the
// libm returns x + x).  Callers that know a bound for several arguments at once test it once (exp32_in_range_3) instead of once
// per call -- the test is a compare and a branch per expf, and the NDProfile kernels call expf fifteen times per point.
RLM_FN float exp32_in_range(float x, const Tables &t);
// true when |a|, |b| and |c| are all below 88 -- NaNs are ignored by the maximum and may pass: see above
RLM_FN bool exp32_in_range_3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)) < 88.0f;
#else
    const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c));
    return m < 88.0f;
#endif
}

RLM_FN float exp32(float x, const Tables &t)
{
    const uint32_t abstop = (f2u(x) >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                              // |x| >= 88 or NaN
        if (f2u(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return u2f(0x7f800000u);                 // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                            // underflow
    }
    return exp32_in_range(x, t);
}

RLM_FN float exp32_in_range(float x, const Tables &t)
{
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N, Shift = 0x1.8p+52;
    const double xd = (double)x;
#if RLM_GLIBC_FMA
    // the FMA build never rounds z = InvLn2N * x on its own: kd and r are fmas of the exact product
    double kd = __builtin_fma(InvLn2N, xd, Shift);
    const uint64_t ki = d2u(kd);
    kd -= Shift;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
#else
    const double z = InvLn2N * xd;
    double kd = z + Shift;
    const uint64_t ki = d2u(kd);
    kd -= Shift;
    const double r = z - kd;
#endif
    return exp2_tail(r, ki, 0x1.c6af84b912394p-5 / N / N / N, 0x1.ebfce50fac4f3p-3 / N / N,
                     0x1.62e42ff0c52d6p-1 / N, t, 0);
}

RLM_FN float log32(float x, const Tables &t)
{
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -u2f(0x7f800000u);                      // log(0) = -inf
        if (ix == 0x7f800000u) return x;                                // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);   // negative or NaN
        ix = f2u(x * 0x1p23f);                                          // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = t.invc[i], logc = t.logc[i];
    const double z = (double)u2f(iz);
    const double r = mad(z, invc, -1.0);
    const double y0 = mad((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = mad(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = mad(-0x1.00ea348b88334p-2, r2, y);
    y = mad(y, r2, y0 + r);
    return (float)y;
}

// 0: y is not an integer, 1: odd integer, 2: even integer (glibc e_powf.c checkint)
RLM_FN int pow_checkint(uint32_t iy)
{
    const int e = (int)(iy >> 23) & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}

// powf (glibc e_powf.c)
RLM_FN float pow32(float x, float y, const Tables &t)
{
    uint32_t ix = f2u(x);
    uint64_t sign_bias = 0;
    const uint32_t iy = f2u(y);
    const bool yspecial = (2u * iy - 1u) >= (2u * 0x7f800000u - 1u);    // y is 0, inf or nan
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || yspecial) {
        if (yspecial) {
            if (2u * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if ((2u * ix - 1u) >= (2u * 0x7f800000u - 1u)) {                // x is 0, inf or nan
            float x2 = x * x;
            if ((ix & 0x80000000u) && pow_checkint(iy) == 1) x2 = -x2;  // -0 / -inf: odd integer y keeps the sign
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {                                         // finite x < 0
            const int yint = pow_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                                         // subnormal x
            ix = f2u(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = t.invc[i], logc = t.log2c[i];
    const double z = (double)u2f(iz);
    const double r = mad(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = mad(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = mad(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = mad(0x1.71547652ab82bp0, r, y0);
    q = mad(p, r2, q);
    yy = mad(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (((d2u(ylogx) >> 47) & 0xffffu) >= (d2u(126.0) >> 47)) {         // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return u2f(sign_bias ? 0xff800000u : 0x7f800000u);
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    // exp2_inline: the argument is already a double, nothing to contract in the reduction
    const double Shift = 0x1.8p+52 / 32.0;
    double kd = ylogx + Shift;
    const uint64_t ki = d2u(kd);
    kd -= Shift;
    const double rr = ylogx - kd;
    return exp2_tail(rr, ki, 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1, t, sign_bias);
}

// a / b when y = RN(1 / b) is at hand (rcp32_w(b), kept per shading point for a denominator that many divisions share):
// q0 = RN(a y) is within 1.5 ulp of a / b; one correction q1 = RN(q0 + RN(a - b q0) y) makes it faithful (within 0.5 ulp +
// 2^-22 ulp), and with a faithful q1 the residual a - b q1 is exact and q2 = RN(q1 + (a - b q1) y) is the correctly
// rounded quotient (Markstein, IBM J. R&D 34 (1990); Muller et al., Handbook of Floating-Point Arithmetic, 2nd ed.,
// theorem 4.9).  Needs every intermediate representable: here for 2^-14 <= |b| <= 2^14 and 2^-75 <= |a| <= 2^40 (the
// residuals are multiples of 2^(eb + eq - 46) >= 2^-149; fp32 subnormals are on).  Five full-rate instructions against
// the fifteen fma-equivalents of the IEEE sequence; the CALLER guarantees the window (b once per point, a by a compare
// where it is not known) -- zero, infinite and NaN operands do NOT come out as IEEE division makes them.
// tools/micro/divy.hip: 64 x 2^30 random pairs inside the window, mantissas near their ends and the window's corners
// included, 0 differences from the compiler's division.
RLM_FN float div32_y(float a, float b, float y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
#else
    (void)y;
    return a / b;
#endif
}

// powf(x, 5) for the Schlick weights (x = a clamped 1 - cos).  The host libm's powf(x, 5) is NOT the correctly rounded
// x^5: its fp64 log2 / exp2 approximations put 145 179 of the 1 065 353 217 arguments in [0, 1] on the other side of an
// fp32 rounding boundary -- but only where the exact x^5 lies within 2^-9.22 ulp of one (tools/micro/pow5.hip, every
// argument).  So: x^5 by three fp64 products; when that lands within 2^-8 ulp of a boundary (0.16 % of the arguments),
// or x is outside [2^-25, 1] (subnormal or zero result, negative, above 1, NaN), the full routine; otherwise the fp32
// rounding of the product IS the libm's result.
RLM_FN float pow5_32(float x, const Tables &t)
{
    const double d = (double)x;
    const double d2 = d * d;
    const double d5 = (d2 * d2) * d;
    const uint32_t low = (uint32_t)d2u(d5) & 0x1fffffffu;      // the 29 bits fp32 drops; the boundary is 2^28
    if (__builtin_expect(!(x >= 0x1p-25f && x <= 1.0f) || low - (0x10000000u - 0x00200000u) <= 0x00400000u, 0))
        return pow32(x, 5.0f, t);
    return (float)d5;
}

// =================================================================================================
// Wave-friendly forms.  Same arithmetic, same results, but the range cases are expressed as
// selects around ONE shared division / polynomial instead of separate branches, because the 64
// lanes of a wavefront land in different ranges and would otherwise execute every branch in turn.
// tests/native/libm_faithful.cpp checks  *_v == plain form == host libm  on every argument.
// =================================================================================================

RLM_FN float sel(bool c, float a, float b) { return c ? a : b; }

// atanf for finite x (NaN propagates through the arithmetic; |x| >= 2^25 handled by a select)
RLM_FN float atan32_v(float x)
{
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u), aT3 = u2f(0xbde38e38u),
                aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u), aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu),
                aT8 = u2f(0x3d4bda59u), aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = fabs32(x);
    const bool r0 = ix < 0x3ee00000;      // |x| < 7/16: no reduction, signed x
    const bool r1 = ix < 0x3f300000;      // < 11/16
    const bool r2 = ix < 0x3f980000;      // < 19/16
    const bool r3 = ix < 0x401c0000;      // < 39/16
    // t = num / den ; r0 uses x / 1 (exact)
    const float num = r0 ? x : r1 ? (2.0f * ax - 1.0f) : r2 ? (ax - 1.0f) : r3 ? (ax - 1.5f) : -1.0f;
    const float den = r0 ? 1.0f : r1 ? (2.0f + ax) : r2 ? (ax + 1.0f) : r3 ? (1.0f + 1.5f * ax) : ax;
    const float hi = r1 ? u2f(0x3eed6338u) : r2 ? u2f(0x3f490fdau) : r3 ? u2f(0x3f7b985eu) : u2f(0x3fc90fdau);
    const float lo = r1 ? u2f(0x31ac3769u) : r2 ? u2f(0x33222168u) : r3 ? u2f(0x33140fb4u) : u2f(0x33a22168u);
    const float t = div32_m(num, den);          // den = 1 or in [1, 2^26]
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float ts = t * (s1 + s2);
    const float small = t - ts;                                   // id < 0
    float big = hi - ((ts - lo) - t);
    big = hx < 0 ? -big : big;
    float res = r0 ? small : big;
    res = (ix < 0x31000000) ? x : res;                            // |x| < 2^-29
    const float huge = u2f(0x3fc90fdau) + u2f(0x33a22168u);       // atanhi[3] + atanlo[3]
    res = (ix >= 0x4c000000 && ix <= 0x7f800000) ? (hx > 0 ? huge : -huge) : res;
    return res;
}

// atan2f.  Infinite or NaN arguments take the branchy routine above (never seen by the closures, so the
// branch is skipped by whole wavefronts).  x == 1 and y == +-0 need no special case: the general path
// gives the same bits (atanf is odd; pi - (0 - pi_lo) rounds to pi).
RLM_FN float atan2_32_v(float y, float x)
{
    const float pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu), pi_lo = u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (__builtin_expect(ix >= 0x7f800000 || iy >= 0x7f800000, 0)) return atan2_32(y, x);
    const int32_t k = (iy - ix) >> 23;
    float z = atan32_v(fabs32(div32(y, x)));
    z = (k > 60) ? (pi_o_2 + 0.5f * pi_lo) : z;
    z = (hx < 0 && k < -60) ? 0.0f : z;
    const float zl = z - pi_lo;
    float res = hx < 0 ? (hy < 0 ? zl - pi : pi - zl) : (hy < 0 ? -z : z);
    // y == +-0: +-0 for x >= 0 (sign of y), +-pi for x < 0
    res = (iy == 0) ? (hx < 0 ? (hy < 0 ? -pi : pi) : y) : res;
    // x == +-0, y != 0
    res = (ix == 0 && iy != 0) ? (hy < 0 ? -pi_o_2 : pi_o_2) : res;
    return res;
}

// The same atanf with the per-range constants read from a table (LDS on the device) instead of selected: the lanes of
// a wavefront fall into all five ranges, and choosing numerator, denominator, atanhi and atanlo took 16 selects on
// top of computing every candidate.  num = A|x| + B and den = C|x| + D are the same operations as the written forms
// (2|x| - 1, |x| - 1.5, 1 + 1.5|x| ...: a product that is exact or the very product of the original, then one
// addition), so the quotient is the same; all 2^32 arguments: tools/libm_exhaustive.py (atan2f(y, 1) slice).
RLM_FN float atan32_t(float x, const Tables &tab)
{
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u), aT3 = u2f(0xbde38e38u),
                aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u), aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu),
                aT8 = u2f(0x3d4bda59u), aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = fabs32(x);
    const bool r0 = ix < 0x3ee00000;      // |x| < 7/16: no reduction, signed x
    const int id = (ix >= 0x3f300000) + (ix >= 0x3f980000) + (ix >= 0x401c0000);
    const float *k = tab.atan_k[id];
    const float A = k[0], B = k[1], C = k[2], D = k[3], hi = k[4], lo = k[5];
    const float num = r0 ? x : A * ax + B;
    const float den = r0 ? 1.0f : C * ax + D;
    const float t = div32_m(num, den);
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float ts = t * (s1 + s2);
    const float small = t - ts;                                   // id < 0
    float big = hi - ((ts - lo) - t);
    big = hx < 0 ? -big : big;
    float res = r0 ? small : big;
    res = (ix < 0x31000000) ? x : res;                            // |x| < 2^-29
    const float huge = u2f(0x3fc90fdau) + u2f(0x33a22168u);       // atanhi[3] + atanlo[3]
    res = (ix >= 0x4c000000 && ix <= 0x7f800000) ? (hx > 0 ? huge : -huge) : res;
    return res;
}

RLM_FN float atan2_32_t(float y, float x, const Tables &tab)
{
    const float pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu), pi_lo = u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (__builtin_expect(ix >= 0x7f800000 || iy >= 0x7f800000, 0)) return atan2_32(y, x);
    const int32_t k = (iy - ix) >> 23;
    float z = atan32_t(fabs32(div32(y, x)), tab);
    z = (k > 60) ? (pi_o_2 + 0.5f * pi_lo) : z;
    z = (hx < 0 && k < -60) ? 0.0f : z;
    const float zl = z - pi_lo;
    float res = hx < 0 ? (hy < 0 ? zl - pi : pi - zl) : (hy < 0 ? -z : z);
    res = (iy == 0) ? (hx < 0 ? (hy < 0 ? -pi : pi) : y) : res;
    res = (ix == 0 && iy != 0) ? (hy < 0 ? -pi_o_2 : pi_o_2) : res;
    return res;
}

// atan2f with the exceptional arguments on a branch of their own.  Of the selects of atan2_32_t / atan32_t seven exist
// for zeros, infinities, NaNs, subnormals and quotients that are tiny or huge (|y / x| < 2^-29: atanf returns its
// argument; >= 2^25: pi/2; exponents more than 60 apart: the atan2f shortcuts) -- compares and selects cost 1.5 x an fma
// each on gfx950 (profiles/r02_valu_rates.txt) and a closure's two atan2f calls never need them.  Both operands normal and
// their exponent fields no more than 27 below / 23 above each other  =>  2^-28 < |y / x| < 2^24, none of the seven fires,
// and what remains is the same expressions in the same order.  Anything else takes atan2_32_t (the wave pays for it).
RLM_FN float atan2_32_q(float y, float x, const Tables &tab)
{
    const float pi = u2f(0x40490fdbu), pi_lo = u2f(0xb3bbbd2eu);
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u), aT3 = u2f(0xbde38e38u),
                aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u), aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu),
                aT8 = u2f(0x3d4bda59u), aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const uint32_t ix = (uint32_t)hx & 0x7fffffffu, iy = (uint32_t)hy & 0x7fffffffu;
    const int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
    if (__builtin_expect(ix - 0x00800000u >= 0x7f000000u || iy - 0x00800000u >= 0x7f000000u || (uint32_t)(k + 27) > 50u, 0))
        return atan2_32_t(y, x, tab);
    const float q = fabs32(div32(y, x));                          // 2^-28 < q < 2^24
    // atan32_t(q) for such q: positive, neither tiny nor huge
    const uint32_t iq = f2u(q);
    const bool r0 = iq < 0x3ee00000u;                             // q < 7/16: no reduction
    const int id = (iq >= 0x3f300000u) + (iq >= 0x3f980000u) + (iq >= 0x401c0000u);
    const float *c = tab.atan_k[id];
    const float A = c[0], B = c[1], C = c[2], D = c[3], hi = c[4], lo = c[5];
    const float num = r0 ? q : A * q + B;
    const float den = r0 ? 1.0f : C * q + D;
    const float t = div32_m(num, den);
    const float zz = t * t;
    const float w = zz * zz;
    const float s1 = zz * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float ts = t * (s1 + s2);
    const float z = r0 ? t - ts : hi - ((ts - lo) - t);
    // the quadrant, as atan2_32_t
    const float zl = z - pi_lo;
    return hx < 0 ? (hy < 0 ? zl - pi : pi - zl) : (hy < 0 ? -z : z);
}

// acosf with |x| < 2^-57, |x| >= 1 and NaN on a branch of their own (acos32): the three trailing selects of acos32_v
// and the division that makes its NaN go away; the rest is acos32_v
RLM_FN float acos32_q(float x)
{
    const float pi = u2f(0x40490fdau), pio2_hi = u2f(0x3fc90fdau), pio2_lo = u2f(0x33a22168u);
    const float pS0 = u2f(0x3e2aaaabu), pS1 = u2f(0xbea6b090u), pS2 = u2f(0x3e4e0aa8u), pS3 = u2f(0xbd241146u),
                pS4 = u2f(0x3a4f7f04u), pS5 = u2f(0x3811ef08u);
    const float qS1 = u2f(0xc019d139u), qS2 = u2f(0x4001572du), qS3 = u2f(0xbf303361u), qS4 = u2f(0x3d9dc62eu);
    const int32_t hx = (int32_t)f2u(x);
    const uint32_t ix = (uint32_t)hx & 0x7fffffffu;
    if (__builtin_expect(ix - 0x23000001u >= 0x3f800000u - 0x23000001u, 0)) return acos32(x);
    const bool mid = ix < 0x3f000000u;                            // |x| < 0.5
    const bool neg = hx < 0;
    const float z = mid ? x * x : (neg ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = div32_m(p, q);              // q in [0.3, 1.1] for z in [0, 1/2]
    float res = pio2_hi - (x - (pio2_lo - x * r));
    if (!mid) {
        const float s = sqrt32<false>(z);         // z = (1 -+ x) / 2 >= 2^-26 here
        const float df = u2f(f2u(s) & 0xfffff000u);
        const float c = div32_m(z - df * df, s + df);   // s + df >= 2^-12
        const float r_pos = 2.0f * (df + (r * s + c));
        const float r_neg = pi - 2.0f * (s + (r * s - pio2_lo));
        res = neg ? r_neg : r_pos;
    }
    return res;
}

// acosf for |x| <= 1 (|x| > 1 gives NaN through the sqrt of a negative number)
RLM_FN float acos32_v(float x)
{
    const float pi = u2f(0x40490fdau), pio2_hi = u2f(0x3fc90fdau), pio2_lo = u2f(0x33a22168u);
    const float pS0 = u2f(0x3e2aaaabu), pS1 = u2f(0xbea6b090u), pS2 = u2f(0x3e4e0aa8u), pS3 = u2f(0xbd241146u),
                pS4 = u2f(0x3a4f7f04u), pS5 = u2f(0x3811ef08u);
    const float qS1 = u2f(0xc019d139u), qS2 = u2f(0x4001572du), qS3 = u2f(0xbf303361u), qS4 = u2f(0x3d9dc62eu);
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const bool mid = ix < 0x3f000000;                             // |x| < 0.5
    const bool neg = hx < 0;
    const float z = mid ? x * x : (neg ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = div32_m(p, q);              // q in [0.3, 1.1] for z in [0, 1/2]
    const float r_mid = pio2_hi - (x - (pio2_lo - x * r));
    float res = r_mid;
    if (!mid) {
        const float s = sqrt32<false>(z);         // z = (1 -+ x) / 2: 0, >= 2^-26, or negative (|x| > 1 -> NaN)
        const float df = u2f(f2u(s) & 0xfffff000u);
        const float c = div32_m(z - df * df, s + df);   // s + df >= 2^-12 (z >= 2^-26) unless |x| = 1, replaced below
        const float r_pos = 2.0f * (df + (r * s + c));
        const float r_neg = pi - 2.0f * (s + (r * s - pio2_lo));
        res = neg ? r_neg : r_pos;
    }
    res = (ix <= 0x23000000) ? pio2_hi + pio2_lo : res;           // |x| < 2^-57
    res = (ix == 0x3f800000) ? (hx > 0 ? 0.0f : pi + 2.0f * pio2_lo) : res;
    res = (ix > 0x3f800000) ? (x - x) / (x - x) : res;
    return res;
}

// fp64 reduction by pi/2 shared by sinf / cosf / tanf: reduce_fast of s_sincosf.h for |y| < 120, reduce_large
// beyond (then the result is |y| reduced, *large_negative tells the caller about the sign; inf / NaN -> NaN).
// FULL = false drops the branch for |y| >= 120: the form the closure kernels use, whose angles are bounded by
// construction (results of atan2f / acosf, or 2 pi xi with xi in [0, 1)); NaN still propagates.  The branch is
// never taken there, but its presence costs 1.7 % of the reflect+refract kernel.
template <bool FULL, bool CONTRACT>
RLM_FN double reduce_pio2(float y, int *np, bool *large_negative)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double x = (double)y;
    *large_negative = false;
    if (FULL && __builtin_expect(((f2u(y) >> 20) & 0x7ffu) >= 0x42fu, 0)) {
        if (((f2u(y) >> 23) & 0xffu) == 0xffu) { *np = 0; return (double)(y - y); }
        *large_negative = (f2u(y) >> 31) != 0;
        return reduce_large(f2u(y), np);
    }
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    // sinf / cosf (FMA build: one fma); tanf's __ieee754_rem_pio2f is built once, uncontracted
    return CONTRACT ? mad(-(double)n, hpi, x) : x - (double)n * hpi;
}

// sinf and cosf of the same argument (one reduction, both polynomials)
template <bool FULL = true>
RLM_FN void sincos32_v(float y, float *sinp, float *cosp)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    int n;
    bool lneg;
    const double x = reduce_pio2<FULL, true>(y, &n, &lneg);   // n == 0 and x == y whenever |y| < pi/4
    const int m = n + (lneg ? 1 : 0);             // s_sinf.c / s_cosf.c: signs from n + sign beyond 120
    // sign[m & 3] of s_sincosf.h = {+, -, -, +}: negative iff bit 1 of m + 1 is set; applied as an xor on the sign bit
    const double xs = u2d(d2u(x) ^ ((uint64_t)(uint32_t)((m + 1) & 2) << 62));
    const double x2 = x * x;
    const double x3 = xs * x2;
    const double s1 = mad(x2, S3, S2);
    const double x5 = x3 * x2;
    const double sv = mad(x3, S1, xs);
    double sres = mad(x5, s1, sv);
    const double x4 = x2 * x2;
    const double c2 = mad(x2, C4, C3);
    const double c1 = mad(x2, C1, C0);
    const double x6 = x4 * x2;
    const double cv = mad(x4, C2, c1);
    double cres = mad(x6, c2, cv);
    cres = u2d(d2u(cres) ^ ((uint64_t)(uint32_t)(m & 2) << 62));   // second table: every cosine coefficient negated
    const float fs = (float)sres, fc = (float)cres;               // round both, then pick (fp32 selects)
    float sf = (n & 1) ? fc : fs;
    float cf = (n & 1) ? fs : fc;
    const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
    if (top < 0x398u) { sf = y; cf = 1.0f; }      // |y| < 2^-12
    *sinp = sf;
    *cosp = cf;
}

// tanf: one fp64 reduction, then k_tanf.c with its single division shared
template <bool FULL = true>
RLM_FN float tan32_v(float xin)
{
    const float pio4 = u2f(0x3f490fdau), pio4lo = u2f(0x33222168u);
    const float T0 = u2f(0x3eaaaaabu), T1 = u2f(0x3e088889u), T2 = u2f(0x3d5d0dd1u), T3 = u2f(0x3cb327a4u),
                T4 = u2f(0x3c11371fu), T5 = u2f(0x3b6b6916u), T6 = u2f(0x3abede48u), T7 = u2f(0x3a1a26c8u),
                T8 = u2f(0x398137b9u), T9 = u2f(0x38a3f445u), T10 = u2f(0x3895c07au), T11 = u2f(0xb79bae5fu),
                T12 = u2f(0x37d95384u);
    // s_tanf.c: |x| <= pi/4 goes to the kernel unreduced; the reduction returns n = 0, y0 = x,
    // y1 = 0 there, so it is applied unconditionally
    int n;
    bool lneg;
    double dx = reduce_pio2<FULL, false>(xin, &n, &lneg);
    if (lneg) { dx = -dx; n = -n; }
    float x = (float)dx;
    float y = (float)(dx - (double)x);
    const bool direct = ((int32_t)f2u(xin) & 0x7fffffff) <= 0x3f490fda;
    x = direct ? xin : x;
    y = direct ? 0.0f : y;
    const int iy = direct ? 1 : 1 - ((n & 1) << 1);
    const float fiy = (float)iy;

    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const float x_in = x;                                         // for the tiny-argument returns
    const bool big = ix >= 0x3f2ca140;                            // |x| >= 0.6744
    const float sgn = (float)(1 - ((hx >> 30) & 2));
    if (big) {
        const float xa = hx < 0 ? -x : x;
        const float ya = hx < 0 ? -y : y;
        const float zz = pio4 - xa;
        const float ww = pio4lo - ya;
        x = zz + ww;
        y = 0.0f;
    }
    const float z = x * x;
    float w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    const float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    const float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    // one division: w*w/(w+iy) in the big case, -1/w when the cotangent is wanted
    const float q = div32_m(big ? w * w : -1.0f, big ? w + fiy : w);   // w + fiy in [0.8, 1.2]; -1 / w is used for |w| >= 1e-8 only
    const float r_big = sgn * (fiy - 2.0f * (x - (q - r)));
    const float zt = u2f(f2u(w) & 0xfffff000u);
    const float vt = r - (zt - x);
    const float t = u2f(f2u(q) & 0xfffff000u);
    const float st = 1.0f + t * zt;
    const float r_cot = t + q * (st + t * vt);
    float res = big ? r_big : (iy == 1 ? w : r_cot);
    // |x| >= 0.6744 and the reflected argument below 2^-13
    res = (big && fabs32(x) < 0x1p-13f) ? sgn * fiy * (1.0f - 2.0f * fiy * x) : res;
    // |x| < 2^-13 on entry to the kernel
    if (((int32_t)f2u(x_in) & 0x7fffffff) < 0x39000000 && (int)x_in == 0) {
        const int32_t ixx = (int32_t)f2u(x_in) & 0x7fffffff;
        res = ((ixx | (iy + 1)) == 0) ? 1.0f / fabs32(x_in) : (iy == 1 ? x_in : -1.0f / x_in);
    }
    return res;
}

// tanf with its two tiny-argument exits on a branch of their own (tan32_v): a reduced argument below 2^-13 (x within
// 2^-13 of a multiple of pi/2; +0 itself excepted: the arithmetic below returns +0 for it, and the closures do pass +0)
// and a reflected argument below 2^-13 (x within 2^-13 of an odd multiple of pi/4).  The rest is tan32_v, same expressions in the same order.
template <bool FULL = true>
RLM_FN float tan32_q(float xin)
{
    const float pio4 = u2f(0x3f490fdau), pio4lo = u2f(0x33222168u);
    const float T0 = u2f(0x3eaaaaabu), T1 = u2f(0x3e088889u), T2 = u2f(0x3d5d0dd1u), T3 = u2f(0x3cb327a4u),
                T4 = u2f(0x3c11371fu), T5 = u2f(0x3b6b6916u), T6 = u2f(0x3abede48u), T7 = u2f(0x3a1a26c8u),
                T8 = u2f(0x398137b9u), T9 = u2f(0x38a3f445u), T10 = u2f(0x3895c07au), T11 = u2f(0xb79bae5fu),
                T12 = u2f(0x37d95384u);
    int n;
    bool lneg;
    double dx = reduce_pio2<FULL, false>(xin, &n, &lneg);
    if (lneg) { dx = -dx; n = -n; }
    float x = (float)dx;
    float y = (float)(dx - (double)x);
    const uint32_t ixin = f2u(xin) & 0x7fffffffu;
    const bool direct = ixin <= 0x3f490fdau;
    x = direct ? xin : x;
    y = direct ? 0.0f : y;
    const int iy = direct ? 1 : 1 - ((n & 1) << 1);
    const float fiy = (float)iy;

    const int32_t hx = (int32_t)f2u(x);
    const uint32_t ix = (uint32_t)hx & 0x7fffffffu;
    const bool big = ix >= 0x3f2ca140u;                           // |x| >= 0.6744
    const float sgn = (float)(1 - ((hx >> 30) & 2));
    const float xa = hx < 0 ? -x : x;
    const float ya = hx < 0 ? -y : y;
    const float xr = (pio4 - xa) + (pio4lo - ya);                 // the reflected argument of the big case
    // (the tiny test is on the REDUCED argument, as in tan32_v; xin = +0 is the one tiny argument the arithmetic below
    // gets right: direct, iy = 1, w = +0)
    if (__builtin_expect((ix < 0x39000000u && f2u(xin) != 0u) || (big && fabs32(xr) < 0x1p-13f), 0))
        return tan32_v<FULL>(xin);
    x = big ? xr : x;
    y = big ? 0.0f : y;
    const float z = x * x;
    float w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    const float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    const float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    const float q = div32_m(big ? w * w : -1.0f, big ? w + fiy : w);
    const float r_big = sgn * (fiy - 2.0f * (x - (q - r)));
    const float zt = u2f(f2u(w) & 0xfffff000u);
    const float vt = r - (zt - x);
    const float t = u2f(f2u(q) & 0xfffff000u);
    const float st = 1.0f + t * zt;
    const float r_cot = t + q * (st + t * vt);
    return big ? r_big : (iy == 1 ? w : r_cot);
}

} // namespace rlm
