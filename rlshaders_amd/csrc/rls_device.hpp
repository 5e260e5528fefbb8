// rls_device.hpp -- device-side closure arithmetic for gfx950 (CDNA4), fp32, one shading point
// per lane.  Hand-written for this library; the reference lines each routine reproduces are
// cited (paths relative to the reference repository).  Build with -ffp-contract=off and without
// fast-math: results must match the reference's CPU closures to <= 1e-5 relative.
//
// Structure (differs from the reference on purpose): every closure is split into
//   * a per-point *setup* that holds everything independent of the random numbers -- frame,
//     alphas, and for visible-normal sampling the whole stretched-view analysis with its
//     atan2f/acosf/tanf/sincosf calls -- and
//   * a per-sample part that is only mul/add/div/sqrt.
// Kernels that draw several samples per point (reflect+refract, rlSkin's two GGX lobes on one
// frame, the n^2-spp integrators) run the setup once.  The split hoists work; it never changes
// the order of floating-point operations inside an expression.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rls_libm.hpp"

namespace rlsd {

// ---- Arnold constants as the reference sees them (public SDK 4.x values) --------------------
constexpr float kPi = 3.14159265f;
constexpr float kTwoPi = 6.28318530f;
constexpr float kHalfPi = 1.57079632f;
constexpr float kInvPi = 0.31830988f;
constexpr float kEps = 1e-4f;   // AI_EPSILON

struct V3 { float x, y, z; };
struct V2 { float x, y; };

#define RLS_DEV __device__ __forceinline__
#ifndef RLS_BLOCK
#define RLS_BLOCK 256     // threads per workgroup of every kernel (rls_internal.hpp, kBlock)
#endif

RLS_DEV float sqr(float a) { return a * a; }
RLS_DEV float absf(float a) { return a < 0.0f ? -a : a; }
RLS_DEV float maxf(float a, float b) { return a > b ? a : b; }
RLS_DEV float minf(float a, float b) { return a < b ? a : b; }
RLS_DEV float clampf(float v, float lo, float hi) { return maxf(lo, minf(v, hi)); }
// LERP(t, a, b) = (1 - t) * a + b * t
RLS_DEV float lerpf(float t, float a, float b) { return ((1.0f - t) * a) + (b * t); }
RLS_DEV float sgnf(float a) { return a < 0.0f ? -1.0f : 1.0f; }

RLS_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RLS_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RLS_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RLS_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
RLS_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RLS_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RLS_DEV V3 cross(V3 a, V3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// AiV3RotateToFrame(a, u, v, w) = a.x*u + a.y*v + a.z*w
RLS_DEV V3 to_frame(V3 a, V3 u, V3 v, V3 w)
{
    return mk(a.x * u.x + a.y * v.x + a.z * w.x,
              a.x * u.y + a.y * v.y + a.z * w.y,
              a.x * u.z + a.y * v.z + a.z * w.z);
}
RLS_DEV bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
RLS_DEV bool is_finite3(V3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

struct Frame { V3 U, V, N; };

// ---- arithmetic policy ---------------------------------------------------------------------------
// RLS_FAST = 0 (the product default, "EXACT"): IEEE division, correctly rounded square root and
// the host-libm-faithful angle functions of rls_libm.hpp -- every closure output that involves no
// exp/log/pow is bit-identical to the CPU closures.
// RLS_FAST = 1 ("FAST", opt-in through rls_context_set_math_mode): v_rcp_f32 / v_sqrt_f32 /
// v_sin_f32 / v_exp_f32 grade arithmetic (about 1 ulp per operation) and the visible-normal view
// analysis by vector algebra instead of the reference's angle round trip.  Within 1e-5 of the CPU
// closures wherever those are well conditioned; see DESIGN.md section 2 for the measured tails.
#ifndef RLS_FAST
#define RLS_FAST 0
#endif
#if RLS_FAST
// (round 4 tried FAST's divisions and square roots at ~0.5 ulp everywhere -- v_rcp_f32 + a Newton step with v_div_fixup_f32,
// v_sqrt_f32 + the +-1 ulp correction: config 2 +9 %, rlSkin +8 %, and not one outlier fewer against the CPU closures,
// profiles/r04_fast_refined_timing.txt; the switch was removed in round 5, the code is in history at 1c59d76)
// the visible-normal slope equations amplify every rounding error: there FAST always spent a Newton
// step on the reciprocal (~0.5 ulp) and the exact sqrt
RLS_DEV float refined_div(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}
#define R_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define R_RCP(b) __builtin_amdgcn_rcpf(b)
#define R_RCPW(b) __builtin_amdgcn_rcpf(b)
#define R_TWO_OVER(den) (2.0f * __builtin_amdgcn_rcpf(den))
#define R_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define R_SQRT1P(y) __builtin_amdgcn_sqrtf(1.0f + (y))
#define R_SQRT1M(t) __builtin_amdgcn_sqrtf(1.0f - (t))
#define R_EXP(x) __expf(x)
#define R_EXP_IN_RANGE(x) __expf(x)
#define R_LOG(x) __logf(x)
#define R_POW(x, y) __powf(x, y)
#define R_POW5(x) __powf(x, 5.0f)
#define R_DIVH(a, b) refined_div(a, b)
#define R_RCPH(b) refined_div(1.0f, b)
#define R_RCPG(b) refined_div(1.0f, b)
#define R_RCPHI(b) __builtin_amdgcn_rcpf(b)
#define R_DIVC(x, C) ((x) * (1.0f / (C)))
#define R_DIVCW(x, C) ((x) * (1.0f / (C)))
#define R_SQRTH(x) rlm::sqrt32(x)
#define R_SQRTH1P(y) rlm::sqrt32_1p(y)
RLS_DEV void t_sincos(float x, float *s, float *c) { *s = __sinf(x); *c = __cosf(x); }
RLS_DEV void t_sincos_any(float x, float *s, float *c) { t_sincos(x, s, c); }
RLS_DEV void stage_libm_tables() {}
#else
#define R_DIV(a, b) rlm::div32((a), (b))
#define R_RCP(b) rlm::rcp32(b)
// 1 / x for x that is 0, inf, NaN or in [2^-126, 2^126] by construction (rls_libm.hpp, rcp32_w): say why at the call
#define R_RCPW(b) rlm::rcp32_w(b)
// 2 / den for den = 1 + sqrtf(...) (Smith G1): den is NaN, inf or in [1, 2^64], 1 / den never subnormal, and doubling
// is exact, so 2 * RN(1 / den) = RN(2 / den); all 2^32 arguments of 2 / (1 + sqrtf(1 + y)) checked in tools/micro/exact1.hip
#define R_TWO_OVER(den) (2.0f * rlm::rcp32_w(den))
#define R_SQRT(x) rlm::sqrt32(x)
#define R_DIVH(a, b) rlm::div32((a), (b))
#define R_RCPH(b) rlm::rcp32_w(b)            // of a square root (normalize_h)
// x / C, C one of the compile-time constants checked in tools/micro/exact1.hip (3, 0.3333, 1 - 0.6666, 0.6666 - 0.3333)
#define R_DIVC(x, C) rlm::div32_const((x), (C), 1.0f / (C))
// ... for an x that is inside 2^-100 .. 2^100 and not zero by construction: no range test (say why at the call)
#define R_DIVCW(x, C) rlm::div32_const_w((x), (C), 1.0f / (C))
#define R_RCPHI(b) rlm::rcp32_hi(b)          // 1 / x for x that is 0, NaN or >= 2^-126 in magnitude by construction
#define R_RCPG(b) rlm::rcp32_hi(b)           // of A^2 - 1: 0 or >= 2^-24 in magnitude (A^2 is near 1 or far from it), unbounded above
#define R_SQRTH(x) rlm::sqrt32(x)
// sqrtf(1 + y): no small-argument guard needed (rls_libm.hpp, sqrt32<false>)
#define R_SQRT1P(y) rlm::sqrt32_1p(y)
#define R_SQRTH1P(y) rlm::sqrt32_1p(y)
// sqrtf(1 - t): no guard either (rls_libm.hpp, sqrt32_1m)
#define R_SQRT1M(t) rlm::sqrt32_1m(t)
// exp / log / pow: the host libm's table-driven fp64 algorithms (rls_libm.hpp).  The tables live
// in LDS (640 B per workgroup; each table is at most one 256-byte bank row, so the per-lane
// lookups are conflict-free); a kernel that evaluates any of the three calls
// stage_libm_tables() once before its loop.
static __device__ __constant__ rlm::Tables c_libm_tables = RLM_TABLES_INIT;
static __shared__ rlm::Tables s_libm_tables;
RLS_DEV void stage_libm_tables()
{
    const uint64_t *src = reinterpret_cast<const uint64_t *>(&c_libm_tables);
    uint64_t *dst = reinterpret_cast<uint64_t *>(&s_libm_tables);
    for (unsigned t = threadIdx.x; t < sizeof(rlm::Tables) / 8; t += RLS_BLOCK) dst[t] = src[t];      // every launch is RLS_BLOCK threads wide
    __syncthreads();
}
#define R_EXP(x) rlm::exp32(x, s_libm_tables)
// expf whose range tests the caller has done for several arguments at once (rls_libm.hpp, exp32_in_range_3)
#define R_EXP_IN_RANGE(x) rlm::exp32_in_range(x, s_libm_tables)
#define R_LOG(x) rlm::log32(x, s_libm_tables)
#define R_POW(x, y) rlm::pow32(x, y, s_libm_tables)
#define R_POW5(x) rlm::pow5_32(x, s_libm_tables)   // powf(x, 5.0f)
// t_sincos / t_tan: angles bounded by construction (results of atan2f / acosf, the concentric-disk mapping, the
// in-kernel sampler) -- the forms without the |x| >= 120 branch (rls_libm.hpp).  t_sincos_any: angles computed from
// caller-supplied random numbers (2 pi xi); full domain, so that even numbers outside [0, 1) give what the CPU gives.
RLS_DEV void t_sincos(float x, float *s, float *c) { rlm::sincos32_v<false>(x, s, c); }
RLS_DEV void t_sincos_any(float x, float *s, float *c) { rlm::sincos32_v<true>(x, s, c); }
RLS_DEV float t_atan2(float y, float x) { return rlm::atan2_32_q(y, x, s_libm_tables); }
RLS_DEV float t_acos(float x) { return rlm::acos32_q(x); }
RLS_DEV float t_tan(float x) { return rlm::tan32_q<false>(x); }
#endif


RLS_DEV float length(V3 a) { return R_SQRT(a.x * a.x + a.y * a.y + a.z * a.z); }
// AiV3Normalize: scale by the reciprocal length, zero vector stays zero
RLS_DEV V3 normalize(V3 a)
{
    float t = length(a);
    if (t != 0.0f) t = R_RCPW(t);            // a square root: 0, inf, NaN or in [2^-75, 2^64]
    return mk(a.x * t, a.y * t, a.z * t);
}
// normalize() for vectors whose direction feeds a sharply peaked function (the microfacet normal
// and the half vector under D(h) at low roughness): FAST keeps these at ~0.5 ulp; EXACT is unchanged
RLS_DEV V3 normalize_h(V3 a)
{
    float t = R_SQRTH(a.x * a.x + a.y * a.y + a.z * a.z);
    if (t != 0.0f) t = R_RCPH(t);
    return mk(a.x * t, a.y * t, a.z * t);
}
RLS_DEV float linearstep(float lo, float hi, float t) { return clampf(R_DIV(t - lo, hi - lo), 0.0f, 1.0f); }

// ---- rlUtil ----------------------------------------------------------------------------------
// src/rlUtil.h:21-29
template <bool BOUNDED_PHI = false>
RLS_DEV V3 spherical_direction(float cosTheta, float phi)
{
    float r = R_SQRT1M(sqr(cosTheta));
    float s, c;
    if (BOUNDED_PHI) t_sincos(phi, &s, &c);
    else t_sincos_any(phi, &s, &c);
    return mk(r * c, r * s, cosTheta);
}
// src/rlUtil.h:31-34
RLS_DEV V3 reflect_direction(V3 i, V3 n) { return n * (2.0f * absf(dot(i, n))) - i; }
// src/rlUtil.h:36-39
RLS_DEV float luminance(float r, float g, float b) { return r * 0.212671f + g * 0.715160f + b * 0.072169f; }
// src/rlUtil.cpp:3-27
RLS_DEV V2 concentric_disk(float rx, float ry)
{
    rx = rx * 2.0f - 1.0f;
    ry = ry * 2.0f - 1.0f;
    V2 out;
    if (rx == 0.0f && ry == 0.0f) {
        out.x = 0.0f; out.y = 0.0f;
        return out;
    }
    // the two cases divide different operands; the lanes of a wavefront take both, so the operands are selected and
    // ONE division serves either case (each lane still divides exactly what its case divides)
    const bool wide = absf(rx) > absf(ry);
    const float q = R_DIV(wide ? kHalfPi * 0.5f * ry : 0.5f * rx, wide ? rx : ry);
    const float r = wide ? rx : ry;
    const float phi = wide ? q : kHalfPi * (1.0f - q);
    float s, c;
    t_sincos(phi, &s, &c);
    out.x = r * c;
    out.y = r * s;
    return out;
}
// cosine hemisphere in a frame: src/rlDisney.cpp:359-365, src/rlSss.h:536-545
RLS_DEV V3 cosine_hemisphere(const Frame &fr, float rx, float ry)
{
    V2 d = concentric_disk(rx, ry);
    float z = R_SQRT(maxf(0.0f, 1.0f - sqr(d.x) - sqr(d.y)));
    return to_frame(mk(d.x, d.y, z), fr.U, fr.V, fr.N);
}

// ---- visible-normal sampling (Heitz & d'Eon), src/rlGgx.cpp:14-99 == src/rlDisney.cpp:416-502
// Per-point part: everything up to (and inside sampleSlope, everything before) the first use of
// the random numbers.
struct VndfView {
    float ax, ay;
    float cosPhi, sinPhi;   // of the stretched view azimuth
    float B, B2, G1, invB;  // tanf(theta) terms (valid when !nearNormal)
    bool nearNormal;        // theta < AI_EPSILON -> uniform slope sample
#if !RLS_FAST
    float yG1;              // RN(1 / G1) for the n^2-spp loops' A = 2 rx / G1 - 1 ; 0: G1 outside div32_y's window
#endif
};

// The view direction in the local frame as the reference obtains it -- independent of the roughness,
// so closures that share (wo, N, T) but differ in alpha (rlSkin's sheen and specular lobes) compute
// it once.  EXACT: sphericalDirection(clamp(N.V), atan2f(V.V, U.V)) (src/rlGgx.cpp:68-72);
// FAST: the dot products themselves.
#if RLS_FAST
RLS_DEV V3 vndf_local(V3 view, const Frame &fr)
{
    return mk(dot(fr.U, view), dot(fr.V, view), clampf(dot(fr.N, view), -1.0f, 1.0f));
}
RLS_DEV VndfView vndf_view_from(V3 local, float ax, float ay)
{
    // FAST mode: the reference goes view -> (theta, phi) -> direction -> stretch -> (theta', phi') with
    // atan2f/cosf/sinf/acosf/atan2f/tanf/cosf/sinf; in exact arithmetic that round trip is the identity
    // on the local view vector, so the stretched direction and the sines/cosines/tangent it needs follow
    // from dot products, one rsqrt and one rcp.
    VndfView w;
    w.ax = ax; w.ay = ay;
    float cz = local.z;
    float sx = local.x * ax;
    float sy = local.y * ay;
    float h2 = sx * sx + sy * sy;
    // cos(theta') of the stretched view decides between the closed-form slopes and the uniform fallback at 1 - 1e-4
    // (src/rlGgx.cpp:75-79), and the two give DIFFERENT (equally valid) samples: formed by AiV3Normalize's own sequence --
    // exact sqrt, correctly rounded reciprocal, product -- it rounds as the reference's does whenever the local view agrees
    // to a few ulp (z' moves by 1e-4 of what they move), so FAST takes the reference's side of the threshold
    float z = cz * rlm::rcp32_w(rlm::sqrt32(h2 + cz * cz));
    float h = R_SQRTH(h2);
    bool flat = !(z < (1.0f - kEps));
    w.cosPhi = (flat || h2 == 0.0f) ? 1.0f : R_DIVH(sx, h);
    w.sinPhi = (flat || h2 == 0.0f) ? 0.0f : R_DIVH(sy, h);
    w.nearNormal = flat;
    float B = flat ? 0.0f : R_DIVH(h, cz);                 // tan(theta') = |(sx,sy)| / cz
    w.B = B;
    w.B2 = sqr(B);
    w.G1 = R_DIVH(2.0f, 1.0f + R_SQRTH1P(w.B2));
    w.invB = R_RCP(B);
    return w;
}

#else
RLS_DEV V3 vndf_local(V3 view, const Frame &fr)
{
    float cosThetaV = clampf(dot(fr.N, view), -1.0f, 1.0f);
    float phiV = t_atan2(dot(fr.V, view), dot(fr.U, view));
    return spherical_direction<true>(cosThetaV, phiV);          // phiV = atan2f(...)
}
RLS_DEV VndfView vndf_view_from(V3 local, float ax, float ay)
{
    VndfView w;
    w.ax = ax; w.ay = ay;
    V3 v = local;
    v.x *= ax;
    v.y *= ay;
    v = normalize(v);

    float theta = 0.0f, phi = 0.0f;
    if (v.z < (1.0f - kEps)) {
        theta = t_acos(v.z);
        phi = t_atan2(v.y, v.x);
    }
    t_sincos(phi, &w.sinPhi, &w.cosPhi);
    w.nearNormal = theta < kEps;
    float B = t_tan(theta);
    w.B = B;
    w.B2 = sqr(B);
    w.G1 = R_TWO_OVER(1.0f + R_SQRT1P(w.B2));
    // B = tanf(theta) with theta = 0 or in [acos(1 - 1e-4), pi] (or NaN): 0 or 8.7e-8 <= |B| <= 2.3e7 -- an fp32
    // angle cannot come closer to pi/2 or pi than that
    w.invB = R_RCPW(B);
    // G1 = 2 / (1 + sqrt(1 + B^2)) is in (0, 1]; below 2^-14 (a stretched view within 1e-4 of the horizon) no reciprocal is kept.
    // Only the n^2-spp loops read it (kernels that do not never compute it)
    w.yG1 = (w.G1 >= 0x1p-14f) ? rlm::rcp32_w(w.G1) : 0.0f;
    return w;
}

#endif

RLS_DEV VndfView vndf_view(V3 view, const Frame &fr, float ax, float ay)
{
    return vndf_view_from(vndf_local(view, fr), ax, ay);
}

// rx / (1 - rx) of the uniform slope sample: a function of rx alone.  For 2^-60 <= rx < 1 the denominator is in
// [2^-24, 1] and the short division of rls_libm.hpp rounds like IEEE division (every rx: tools/micro/exact1.hip)
RLS_DEV float ratio_to_one_minus(float rx)
{
#if RLS_FAST
    return R_DIV(rx, 1.0f - rx);
#else
    if (__builtin_expect(!(rx >= 0x1p-60f && rx < 1.0f), 0)) return rx / (1.0f - rx);
    return rlm::div32_m(rx, 1.0f - rx);
#endif
}

// uniformSample lambda, src/rlGgx.cpp:18-25
RLS_DEV V2 uniform_slope(float rx, float ry)
{
    float r = R_SQRT(ratio_to_one_minus(rx));
    float phi = kTwoPi * ry;
    float s, c;
    t_sincos_any(phi, &s, &c);
    V2 slope;
    slope.x = r * c;
    slope.y = r * s;
    return slope;
}

// the rational fit of the inverse CDF in the second slope (src/rlGgx.cpp:52-56), u = 2 |ry - 1/2|
RLS_DEV float slope_y_ratio(float u)
{
    const float num = u * (u * (u * 0.27385f - 0.73369f) + 0.46341f);
    const float den = u * (u * (u * 0.093073f + 0.309420f) - 1.0f) + 0.597999f;
#if RLS_FAST
    return R_DIVH(num, den);
#else
    // for u in [0, 1] (ry in [0, 1]) the denominator falls from 0.598 to 4.9e-4: the short division of rls_libm.hpp
    // rounds like IEEE division for every such u (all 2^32 bit patterns of ry: tools/micro/exact1.hip); other u
    // (random numbers outside [0, 1], NaN) take the IEEE sequence
    if (__builtin_expect(!(u <= 1.0f), 0)) return num / den;
    return rlm::div32_m(num, den);
#endif
}

// Per-sample part of sampleSlope + evalSample: src/rlGgx.cpp:36-60, 89-98.
// vndf_slope_closed: the closed-form slopes (36-60); returns true where the reference takes the uniform fallback
// instead (27: theta < eps, 38: |A^2 - 1| < eps) -- the slopes it writes are then unused.
// LOOP_RECIP (the n^2-spp loops, whose rx comes from the in-kernel sampler: zero or a multiple of 2^-24 below 1, possibly
// divided by the lobe weight -- never in (0, 2^-75)): the caller has checked that w.yG1 != 0 in every active lane and the
// quotient by the per-point G1 goes through its reciprocal (rlm::div32_y), the same correctly rounded value.
template <bool LOOP_RECIP = false>
RLS_DEV bool vndf_slope_closed(const VndfView &w, float rx, float ry, V2 &slope)
{
#if !RLS_FAST
    float A = (LOOP_RECIP ? rlm::div32_y(2.0f * rx, w.G1, w.yG1) : R_DIVH(2.0f * rx, w.G1)) - 1.0f;
#else
    float A = R_DIVH(2.0f * rx, w.G1) - 1.0f;
#endif
    float A2 = sqr(A);
    float tmp = R_RCPG(A2 - 1.0f);
    float D = R_SQRTH(maxf(0.0f, w.B2 * sqr(tmp) - (A2 - w.B2) * tmp));
    float slopeX1 = w.B * tmp - D;
    float slopeX2 = w.B * tmp + D;
    slope.x = (A < 0.0f || slopeX2 > w.invB) ? slopeX1 : slopeX2;

    float sign = 1.0f;
    float u;
    if (ry > 0.5f) {
        u = 2.0f * (ry - 0.5f);
    } else {
        sign = -1.0f;
        u = 2.0f * (0.5f - ry);
    }
    float z = slope_y_ratio(u);
    slope.y = sign * z * R_SQRTH1P(sqr(slope.x));
    return w.nearNormal || absf(A2 - 1.0f) < kEps;
}

// rotate by the view azimuth, unstretch, to world, normalise: src/rlGgx.cpp:89-98
RLS_DEV V3 vndf_from_slope(const VndfView &w, const Frame &fr, V2 slope)
{
    V3 omega;
    omega.x = -(w.cosPhi * slope.x - w.sinPhi * slope.y) * w.ax;
    omega.y = -(w.sinPhi * slope.x + w.cosPhi * slope.y) * w.ay;
    omega.z = 1.0f;
    return normalize_h(to_frame(omega, fr.U, fr.V, fr.N));
}

RLS_DEV V3 vndf_microfacet(const VndfView &w, const Frame &fr, float rx, float ry)
{
    V2 slope;
    if (w.nearNormal) {           // whole wavefronts of near-normal points (low roughness) skip the closed form
        slope = uniform_slope(rx, ry);
    } else if (vndf_slope_closed(w, rx, ry, slope)) {
        slope = uniform_slope(rx, ry);
    }
    return vndf_from_slope(w, fr, slope);
}

// Two microfacet samples per lane (reflect + refract on one closure; the two lobes of rlSkin) with ONE pass of
// the uniform fallback for the whole wavefront.  A few lanes per wavefront need the fallback (near-normal stretched
// view: ~7 % of the points of the mixed-roughness workload) and drag all 64 through uniform_slope's exact division,
// square root and fp64 sincosf -- once per sample.  Here the (rx, ry) pairs of the lanes that need it are packed into
// the low lanes with ds_permute, evaluated once, and handed back with ds_bpermute: the same function on the same
// arguments, on another lane.  Falls back to the per-sample form when the wavefront is not fully active or more
// than 63 evaluations are wanted.
RLS_DEV void vndf_microfacet_pair(const VndfView &w1, const Frame &fr1, float rx1, float ry1,
                                  const VndfView &w2, const Frame &fr2, float rx2, float ry2, V3 &M1, V3 &M2)
{
#if RLS_FAST
    M1 = vndf_microfacet(w1, fr1, rx1, ry1);
    M2 = vndf_microfacet(w2, fr2, rx2, ry2);
#else
    if (__builtin_amdgcn_ballot_w64(true) != ~0ull) {
        M1 = vndf_microfacet(w1, fr1, rx1, ry1);
        M2 = vndf_microfacet(w2, fr2, rx2, ry2);
        return;
    }
    V2 s1, s2;
    const bool n1 = vndf_slope_closed(w1, rx1, ry1, s1);
    const bool n2 = vndf_slope_closed(w2, rx2, ry2, s2);
    const uint64_t m1 = __builtin_amdgcn_ballot_w64(n1), m2 = __builtin_amdgcn_ballot_w64(n2);
    const int c1 = __builtin_popcountll(m1), tot = c1 + __builtin_popcountll(m2);
    if (tot != 0) {
        if (tot <= 63) {
            const int lane = (int)(threadIdx.x & 63u);
            const int k1 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
            const int k2 = c1 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
            // lanes without a request push to lane 63, which holds no request (tot <= 63)
            const int d1 = (n1 ? k1 : 63) * 4, d2 = (n2 ? k2 : 63) * 4;
            const int ax = __builtin_amdgcn_ds_permute(d1, (int)__float_as_uint(rx1));
            const int ay = __builtin_amdgcn_ds_permute(d1, (int)__float_as_uint(ry1));
            const int bx = __builtin_amdgcn_ds_permute(d2, (int)__float_as_uint(rx2));
            const int by = __builtin_amdgcn_ds_permute(d2, (int)__float_as_uint(ry2));
            const float qx = __uint_as_float((uint32_t)(lane < c1 ? ax : bx));
            const float qy = __uint_as_float((uint32_t)(lane < c1 ? ay : by));
            V2 u = { 0.0f, 0.0f };
            if (lane < tot) u = uniform_slope(qx, qy);
            const float u1x = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(k1 * 4, (int)__float_as_uint(u.x)));
            const float u1y = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(k1 * 4, (int)__float_as_uint(u.y)));
            const float u2x = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(k2 * 4, (int)__float_as_uint(u.x)));
            const float u2y = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(k2 * 4, (int)__float_as_uint(u.y)));
            if (n1) { s1.x = u1x; s1.y = u1y; }
            if (n2) { s2.x = u2x; s2.y = u2y; }
        } else {
            if (n1) s1 = uniform_slope(rx1, ry1);
            if (n2) s2 = uniform_slope(rx2, ry2);
        }
    }
    M1 = vndf_from_slope(w1, fr1, s1);
    M2 = vndf_from_slope(w2, fr2, s2);
#endif
}

// A value every lane of the wavefront holds (computed from kernel arguments only) moved to a scalar register: it costs no
// vector register across the tile loop and the branches on it are scalar branches
RLS_DEV float wave_uniform(float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); }
RLS_DEV int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// ---- rlGgx closure state, src/rlGgx.h:130-156 ---------------------------------------------------
struct Ggx {
    Frame fr;
    V3 view;
    float ksR, ksG, ksB;
    float rough;          // mRoughness = max(1e-5, r^2)
    float ax, ay;
    float iorIn, iorOut;
    // values the reference recomputes in every call, hoisted (same expressions, same bits):
    float eta2;           // SQR(mIorOut / mIorIn), src/rlGgx.h:258
    float etaIO;          // mIorIn / mIorOut (the eta of the refraction)
    float vn;             // dot(mViewDir, mAxisN)
    float g1v;            // G1(mViewDir, m, n) where it is not zero, src/rlGgx.h:353-356
};

// The half of the constructor that depends on the node parameters only (src/rlGgx.h:130-156): a kernel whose parameters are
// one value for the batch evaluates it once per thread (ggx_material + ggx_material_wave_uniform), every other kernel per
// point through ggx_make() -- the same expressions either way.
struct GgxMaterial {
    float rough, ax, ay;
    float out, rout;      // max(ior, 1e-4) and its reciprocal
};

// ISOTROPIC: the caller passes anisotropic = 0 (rlSkin's lobes, src/rlSkin.cpp:192,215): aspect = sqrtf(1 - 0 * 0.9) is
// exactly 1, r^2 / 1 and r^2 * 1 are r^2 -- the square root, the division and the product are skipped, same bits
template <bool ISOTROPIC = false>
RLS_DEV GgxMaterial ggx_material(float ior, float roughness, float anisotropic)
{
    GgxMaterial m;
    m.out = maxf(ior, 1e-4f);
    if (ISOTROPIC) {
        m.ax = maxf(1e-4f, sqr(roughness));
        m.ay = m.ax;
    } else {
        float aspect = R_SQRT1M(anisotropic * 0.9f);
        m.ax = maxf(1e-4f, R_DIV(sqr(roughness), aspect));
        m.ay = maxf(1e-4f, sqr(roughness) * aspect);
    }
    m.rough = maxf(1e-5f, sqr(roughness));
    // mIorOut / mIorIn (src/rlGgx.h:258) and mIorIn / mIorOut (refraction): one of the two iors is exactly 1, so one
    // ratio is the other ior itself (x / 1 = x exactly) and the other its reciprocal; out >= 1e-4 (or it is 1e-4)
    m.rout = R_RCPHI(m.out);
    return m;
}

// ks: specColor enters no arithmetic of the constructor (it scales evalBrdf's result), so it may be a per-point plane
// beside a hoisted material
RLS_DEV Ggx ggx_from_material(const GgxMaterial &m, V3 wo, V3 N, V3 T, bool exiting, float ksR, float ksG, float ksB)
{
    Ggx g;
    g.ksR = ksR; g.ksG = ksG; g.ksB = ksB;
    const float in = 1.0f;
    g.iorIn = exiting ? m.out : in;
    g.iorOut = exiting ? in : m.out;
    g.view = wo;
    g.fr.N = N;
    g.fr.U = T;
    g.fr.V = cross(N, T);
    g.ax = m.ax;
    g.ay = m.ay;
    g.rough = m.rough;
    g.eta2 = sqr(exiting ? m.rout : m.out);
    g.etaIO = exiting ? m.out : m.rout;
    g.vn = dot(wo, N);
    {
        float cosSqr = sqr(g.vn);
        float tanSqr = R_RCP(cosSqr) - 1.0f;
        g.g1v = R_TWO_OVER(1.0f + R_SQRT1P(sqr(g.rough) * tanSqr));
    }
    return g;
}

RLS_DEV GgxMaterial ggx_material_wave_uniform(GgxMaterial m)
{
    m.rough = wave_uniform(m.rough); m.ax = wave_uniform(m.ax); m.ay = wave_uniform(m.ay);
    m.out = wave_uniform(m.out); m.rout = wave_uniform(m.rout);
    return m;
}

template <bool ISOTROPIC = false>
RLS_DEV Ggx ggx_make(V3 wo, V3 N, V3 T, bool exiting, float ksR, float ksG, float ksB,
                     float ior, float roughness, float anisotropic)
{
    return ggx_from_material(ggx_material<ISOTROPIC>(ior, roughness, anisotropic), wo, N, T, exiting, ksR, ksG, ksB);
}

// src/rlGgx.h:249-270
RLS_DEV float ggx_fresnel(const Ggx &g, V3 i, V3 m)
{
    float c = absf(dot(i, m));
    float gSqr = g.eta2 - 1.0f + c * c;
    if (gSqr < 0.0f) return 1.0f;
    float gg = R_SQRT(gSqr);
    float gmc = gg - c;
    float gpc = gg + c;
    return 0.5f * sqr(R_DIV(gmc, gpc)) * (1.0f + sqr(R_DIV(c * gpc - 1.0f, c * gmc + 1.0f)));
}

// src/rlGgx.h:332-340
RLS_DEV float ggx_D(const Ggx &g, V3 m)
{
    float mu = dot(m, g.fr.U);
    float mv = dot(m, g.fr.V);
    float mn2 = sqr(dot(g.fr.N, m));
    float den = g.ax * g.ay * sqr(sqr(R_DIVH(mu, g.ax)) + sqr(R_DIVH(mv, g.ay)) + mn2);
    return R_DIV(kInvPi, den);
}

// src/rlGgx.h:343-357
RLS_DEV float ggx_G1(const Ggx &g, V3 v, V3 m, V3 n)
{
    float vm = dot(v, m);
    float vn = dot(v, n);
    if (vm * vn < 0.0f) return 0.0f;
    float cosSqr = sqr(vn);
    float tanSqr = R_RCP(cosSqr) - 1.0f;
    float den = 1.0f + R_SQRT1P(sqr(g.rough) * tanSqr);
    return R_TWO_OVER(den);
}

// G1(mViewDir, m, mAxisN): the value depends on the view only, the zero test on m
RLS_DEV float ggx_G1_view(const Ggx &g, V3 m) { return dot(g.view, m) * g.vn < 0.0f ? 0.0f : g.g1v; }

// src/rlGgx.h:272-275 (i is always mViewDir at the reference's call sites: 300,312)
RLS_DEV float ggx_G(const Ggx &g, V3 o, V3 m) { return ggx_G1_view(g, m) * ggx_G1(g, o, m, g.fr.N); }

// evalBrdf (src/rlGgx.h:110-119,158-165) with reflection() (304-313), and evalPdf (121-127) with
// VNDFKernel::evalPdf (72-80), sharing what both compute: the half vector (normalize(L + V) and
// normalize(V + L) are the same bits; hr = sgn * H only flips signs) and D (even in m).
template <bool WANT_F, bool WANT_PDF>
RLS_DEV void ggx_eval_pdf(const Ggx &g, V3 L, float &fr, float &fg, float &fb, float &pdf)
{
    V3 H = normalize_h(L + g.view);
    float d = ggx_D(g, H);
    if (WANT_PDF) {
        float p = R_DIV(d * ggx_G1_view(g, H), absf(g.vn)) * 0.25f;
        pdf = maxf(p, kEps);
    }
    if (WANT_F) {
        bool small = absf(g.ksR) < kEps && absf(g.ksG) < kEps && absf(g.ksB) < kEps;
        if (is_zero(L) || small) {
            fr = 0.0f; fg = 0.0f; fb = 0.0f;
            return;
        }
        V3 hr = H * sgnf(g.vn);
        float rw = ggx_fresnel(g, g.view, hr);
        float ln = absf(dot(L, g.fr.N));
        float vn = absf(g.vn);
        float refl = R_DIV(rw * ggx_G(g, L, hr) * d * 0.25f, ln * vn);
        float lns = dot(L, g.fr.N);
        fr = g.ksR * refl * lns;
        fg = g.ksG * refl * lns;
        fb = g.ksB * refl * lns;
    }
}

RLS_DEV void ggx_eval(const Ggx &g, V3 L, float &fr, float &fg, float &fb)
{
    float unused;
    ggx_eval_pdf<true, false>(g, L, fr, fg, fb, unused);
}

RLS_DEV float ggx_pdf(const Ggx &g, V3 L)
{
    float a, b, c, pdf;
    ggx_eval_pdf<false, true>(g, L, a, b, c, pdf);
    return pdf;
}

// NDFKernel, src/rlGgx.h:33-50 (alternate, not selected by the reference)
RLS_DEV V3 ndf_microfacet(const Ggx &g, float rx, float ry)
{
    float gg = R_SQRT(R_DIV(rx, 1.0f - rx));
    float phi = kTwoPi * ry;
    float s, c;
    t_sincos_any(phi, &s, &c);
    V3 omega = mk(gg * g.ax * c, gg * g.ay * s, 1.0f);
    return normalize(to_frame(omega, g.fr.U, g.fr.V, g.fr.N));
}
RLS_DEV float ndf_pdf(const Ggx &g, V3 i, V3 m)
{
    float im = absf(dot(i, m));
    float mn = absf(dot(m, g.fr.N));
    return R_DIV(ggx_D(g, m) * mn * 0.25f, im);
}

// getSampleWeight, src/rlGgx.h:294-301
RLS_DEV float ggx_sample_weight(const Ggx &g, V3 i, V3 o, V3 m)
{
    float ih = dot(i, m);
    float mn = absf(dot(m, g.fr.N));
    float in = absf(dot(i, g.fr.N));
    return ggx_G(g, o, m) * absf(R_DIV(ih, in * mn));
}

// Refraction of sg->Rd = -view about the microfacet m, eta = iorIn/iorOut (Walter et al. EGSR'07
// eq. 40); mirror on total internal reflection.  Stands in for the closed AiRefractRay /
// AiReflectRay the reference calls at src/rlGgx.h:230-236.
RLS_DEV bool ggx_refract(const Ggx &g, V3 m, V3 &dir)
{
    V3 i = g.view;
    float eta = g.etaIO;
    float c = dot(i, m);
    float k = 1.0f - eta * eta * (1.0f - c * c);
    bool refracted = !(k < 0.0f);
    if (refracted) {
        float s = sgnf(dot(i, g.fr.N));
        float t = eta * c - s * R_SQRT(k);
        dir = m * t - i * eta;
    } else {
        dir = m * (2.0f * c) - i;
    }
    return refracted;
}

// ---- rlDisney closure state, src/rlDisney.cpp:155-192 -------------------------------------------
struct Disney {
    Frame fr;
    V3 view;
    float f0R, f0G, f0B;          // mSpecularF0
    float shR, shG, shB;          // mSheenColor
    float baseR, baseG, baseB;
    float roughness, subsurface, metallic, clearcoat, clearcoatGloss;
    float specRough;              // mSpecularRoughness
    float ax, ay;
    // Per-point values of expressions the reference re-evaluates in every evalBrdf / evalPdf / evalSample call
    // (same expressions on the same operands, so the same bits), filled by disney_prepare():
    float vn;                     // dot(mViewDir, mAxisN)
    float FV;                     // powf(clamp(1 - vn), 5), src/rlDisney.cpp:221
    float gsV, grV;               // smithG_GGX(vn, mSpecularRoughness), smithG_GGX(vn, 0.25), src/rlDisney.cpp:345,350
    float ccA2m1, ccLogA2;        // D_GTR1: a2 - 1 and logf(a2), a = LERP(gloss, .1, .001), src/rlDisney.cpp:547-549
    float ccw, vnc;               // evalSpecularPdf: clearcoat / (clearcoat + 1), max(1e-4, vn), src/rlDisney.cpp:529,533
    float gtr2Weight;             // sampleSpecularDirection: 1 / (clearcoat + 1), src/rlDisney.cpp:371
    float om;                     // 1 - metallic
#if !RLS_FAST
    float yax, yay;               // RN(1 / ax), RN(1 / ay) for D_GTR2Aniso's two quotients; 0: outside div32_y's window
    float yW, y1mW;               // RN(1 / gtr2Weight), RN(1 / (1 - gtr2Weight)) for the lobe pick's rescaled rx; 0: no reciprocal
#endif
};

// The constructor (src/rlDisney.cpp:155-192) in two halves: what the ten scalars alone decide, and what needs base_color.
// A kernel whose scalars are one value for the batch runs the first half once per thread (and the second as well when
// base_color is uniform too); disney_make() is the two in sequence.
// s: subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint,
//    clearcoat, clearcoat_gloss
struct DisneyTints { float specular, specularTint, sheen, sheenTint; };     // the scalars the base_color half reads
RLS_DEV DisneyTints disney_make_scalars(Disney &d, const float (&s)[10])
{
    DisneyTints t;
    d.subsurface = s[0];
    d.metallic = s[1];
    t.specular = s[2] * 0.08f;
    t.specularTint = s[3];
    d.roughness = s[4];
    float anisotropic = s[5];
    t.sheen = s[6];
    t.sheenTint = s[7];
    d.clearcoat = s[8] * 0.25f;
    d.clearcoatGloss = s[9];
    float aspect = R_SQRT1M(anisotropic * 0.9f);
    d.ax = maxf(1e-2f, R_DIV(sqr(d.roughness), aspect));
    d.ay = maxf(1e-2f, sqr(d.roughness) * aspect);
    d.specRough = sqr(d.roughness);
#if !RLS_FAST
    d.yax = 0.0f; d.yay = 0.0f;      // "no reciprocals": D_GTR2Aniso divides the IEEE way until disney_prepare_material() has run
    d.yW = 0.0f; d.y1mW = 0.0f;
#endif
    return t;
}
RLS_DEV void disney_make_base(Disney &d, const DisneyTints &t, float bR, float bG, float bB)
{
    d.baseR = bR; d.baseG = bG; d.baseB = bB;
    float lum = luminance(bR, bG, bB);
    float tR = 1.0f, tG = 1.0f, tB = 1.0f;
    if (lum > 0.0f) { tR = R_DIV(bR, lum); tG = R_DIV(bG, lum); tB = R_DIV(bB, lum); }
    float mR = lerpf(t.specularTint, 1.0f, tR) * t.specular;
    float mG = lerpf(t.specularTint, 1.0f, tG) * t.specular;
    float mB = lerpf(t.specularTint, 1.0f, tB) * t.specular;
    d.f0R = lerpf(d.metallic, mR, bR);
    d.f0G = lerpf(d.metallic, mG, bG);
    d.f0B = lerpf(d.metallic, mB, bB);
    d.shR = lerpf(t.sheenTint, 1.0f, tR) * t.sheen;
    d.shG = lerpf(t.sheenTint, 1.0f, tG) * t.sheen;
    d.shB = lerpf(t.sheenTint, 1.0f, tB) * t.sheen;
}
RLS_DEV Disney disney_make(V3 wo, V3 N, V3 T, float bR, float bG, float bB, const float (&s)[10])
{
    Disney d;
    d.view = wo;
    d.fr.N = N;
    d.fr.U = T;
    d.fr.V = cross(N, T);
    const DisneyTints t = disney_make_scalars(d, s);
    disney_make_base(d, t, bR, bG, bB);
    return d;
}

// src/rlDisney.cpp:570-577
RLS_DEV float smithG_GGX(float ndv, float alphaG)
{
    float a = alphaG * alphaG;
    float b = ndv * ndv;
    return R_RCP(ndv + R_SQRT(a + b - a * b));
}
// src/rlDisney.cpp:545-551
RLS_DEV float D_GTR1(const Disney &d, float mn2)
{
    float alpha = lerpf(d.clearcoatGloss, 0.1f, 0.001f);
    float a2 = sqr(alpha);
    float den = R_LOG(a2) * (1.0f + (a2 - 1.0f) * mn2);
    return R_DIV((a2 - 1.0f) * kInvPi, den);
}
// src/rlDisney.cpp:561-568
// Round 4: the two quotients by alpha_x, alpha_y -- per-point denominators that every sample of the n^2-spp
// loops divides by -- through their correctly rounded reciprocals (rlm::div32_y: five instructions each instead of the IEEE
// sequence's fifteen fma-equivalents).  div32_y wants 2^-14 <= alpha <= 2^14 (d.yax != 0 says so) and a numerator that is
// zero or in [2^-75, 2^40]: h.u, h.v of a unit half vector are; below 2^-75 the quotient's square is below 2^-122 beside
// the other two terms' >= 1e-2, so its last bit cannot reach the sum; above 2^40, infinite or NaN (hostile inputs) the whole
// wavefront takes the IEEE form -- one test for both quotients.
RLS_DEV float D_GTR2Aniso(const Disney &d, V3 m, float mn2)
{
    float hu = dot(m, d.fr.U);
    float hv = dot(m, d.fr.V);
#if !RLS_FAST
    float qu, qv;
    const bool plain = !(absf(hu) <= 0x1p40f && absf(hv) <= 0x1p40f) || d.yax == 0.0f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(plain) != 0ull, 0)) {
        qu = R_DIV(hu, d.ax);
        qv = R_DIV(hv, d.ay);
    } else {
        qu = rlm::div32_y(hu, d.ax, d.yax);
        qv = rlm::div32_y(hv, d.ay, d.yay);
    }
    float den = d.ax * d.ay * sqr(sqr(qu) + sqr(qv) + mn2);
#else
    float den = d.ax * d.ay * sqr(sqr(R_DIV(hu, d.ax)) + sqr(R_DIV(hv, d.ay)) + mn2);
#endif
    return R_DIV(kInvPi, den);
}

// D_GTR1 with the per-point part (alpha, a2, logf(a2)) taken from disney_prepare(): the same operations
RLS_DEV float D_GTR1_prepared(const Disney &d, float mn2)
{
    float den = d.ccLogA2 * (1.0f + d.ccA2m1 * mn2);
    return R_DIV(d.ccA2m1 * kInvPi, den);
}

// Everything the per-sample verbs compute from the closure alone, in two halves: what depends on the node parameters only
// (a kernel whose parameters are one value for the batch evaluates disney_make() and this half once per thread, then
// disney_wave_uniform()) and what depends on the view.  Needs the libm tables (powf, logf): kernels call
// stage_libm_tables() first.
RLS_DEV void disney_prepare_material(Disney &d)
{
    float alpha = lerpf(d.clearcoatGloss, 0.1f, 0.001f);
    float a2 = sqr(alpha);
    d.ccA2m1 = a2 - 1.0f;
    d.ccLogA2 = R_LOG(a2);
    d.ccw = R_DIV(d.clearcoat, d.clearcoat + 1.0f);
    d.gtr2Weight = R_RCP(d.clearcoat + 1.0f);
    d.om = 1.0f - d.metallic;
#if !RLS_FAST
    // ax, ay = max(1e-2, r^2 / aspect), max(1e-2, r^2 aspect): inside div32_y's window [2^-14, 2^14] unless the roughness is absurd
    const bool win = d.ax <= 0x1p14f && d.ay <= 0x1p14f && d.ax >= 0x1p-14f && d.ay >= 0x1p-14f;
    d.yax = win ? rlm::rcp32_w(d.ax) : 0.0f;
    d.yay = win ? rlm::rcp32_w(d.ay) : 0.0f;
    // gtr2Weight = 1 / (clearcoat + 1) and its complement: in the window for every clearcoat in [2.5e-4, 4 x 16383] or so
    const float w1 = d.gtr2Weight, w2 = 1.0f - d.gtr2Weight;
    d.yW = (w1 >= 0x1p-14f && w1 <= 0x1p14f) ? rlm::rcp32_w(w1) : 0.0f;
    d.y1mW = (w2 >= 0x1p-14f && w2 <= 0x1p14f) ? rlm::rcp32_w(w2) : 0.0f;
#endif
}
RLS_DEV void disney_prepare_view(Disney &d)
{
    d.vn = dot(d.view, d.fr.N);
    d.FV = R_POW5(clampf(1.0f - d.vn, 0.0f, 1.0f));
    d.gsV = smithG_GGX(d.vn, d.specRough);
    d.grV = smithG_GGX(d.vn, 0.25f);
    d.vnc = maxf(1e-4f, d.vn);
}
RLS_DEV void disney_prepare(Disney &d)
{
    disney_prepare_view(d);
    disney_prepare_material(d);
}
// the parameter-only members moved to scalar registers (the frame, the view and disney_prepare_view()'s members are per point)
RLS_DEV void disney_wave_uniform(Disney &d)
{
    d.f0R = wave_uniform(d.f0R); d.f0G = wave_uniform(d.f0G); d.f0B = wave_uniform(d.f0B);
    d.shR = wave_uniform(d.shR); d.shG = wave_uniform(d.shG); d.shB = wave_uniform(d.shB);
    d.baseR = wave_uniform(d.baseR); d.baseG = wave_uniform(d.baseG); d.baseB = wave_uniform(d.baseB);
    d.roughness = wave_uniform(d.roughness); d.subsurface = wave_uniform(d.subsurface); d.metallic = wave_uniform(d.metallic);
    d.clearcoat = wave_uniform(d.clearcoat); d.clearcoatGloss = wave_uniform(d.clearcoatGloss);
    d.specRough = wave_uniform(d.specRough); d.ax = wave_uniform(d.ax); d.ay = wave_uniform(d.ay);
    d.ccA2m1 = wave_uniform(d.ccA2m1); d.ccLogA2 = wave_uniform(d.ccLogA2); d.ccw = wave_uniform(d.ccw);
    d.gtr2Weight = wave_uniform(d.gtr2Weight); d.om = wave_uniform(d.om);
#if !RLS_FAST
    d.yax = wave_uniform(d.yax); d.yay = wave_uniform(d.yay);
    d.yW = wave_uniform(d.yW); d.y1mW = wave_uniform(d.y1mW);
#endif
}
RLS_DEV DisneyTints disney_wave_uniform(DisneyTints t)
{
    t.specular = wave_uniform(t.specular); t.specularTint = wave_uniform(t.specularTint);
    t.sheen = wave_uniform(t.sheen); t.sheenTint = wave_uniform(t.sheenTint);
    return t;
}

// evalDiffuse, src/rlDisney.cpp:199-236 (BRDF without the cosine)
RLS_DEV void disney_eval_diffuse(const Disney &d, V3 L, float &r, float &g, float &b)
{
    r = 0.0f; g = 0.0f; b = 0.0f;
    float ln = dot(L, d.fr.N);
    float vn = dot(d.view, d.fr.N);
    if (ln < kEps || vn < kEps) return;
    V3 H = normalize(L + d.view);
    float lh = dot(L, H);
    float vh = dot(d.view, H);     // the reference's "NdotH" (line 210)
    if (vh < kEps || lh < kEps) return;
    float lh2 = sqr(lh);
    float FL = R_POW5(clampf(1.0f - ln, 0.0f, 1.0f));
    float FV = R_POW5(clampf(1.0f - vn, 0.0f, 1.0f));
    float F90 = 0.5f + 2.0f * d.roughness * lh2;
    float diffuseFactor = lerpf(FL, 1.0f, F90) * lerpf(FV, 1.0f, F90);
    float Fss90 = d.roughness * lh2;
    float Fss = lerpf(FL, 1.0f, Fss90) * lerpf(FV, 1.0f, Fss90);
    float ssFactor = 1.25f * (Fss * (R_RCP(ln + vn) - 0.5f) + 0.5f);
    float mix = lerpf(d.subsurface, diffuseFactor, ssFactor);
    float om = 1.0f - d.metallic;
    r = d.baseR * kInvPi * mix * om;
    g = d.baseG * kInvPi * mix * om;
    b = d.baseB * kInvPi * mix * om;
}

// evalSpecular, src/rlDisney.cpp:318-356
RLS_DEV void disney_eval_specular(const Disney &d, V3 L, float &r, float &g, float &b)
{
    r = 0.0f; g = 0.0f; b = 0.0f;
    float ln = dot(L, d.fr.N);
    float vn = dot(d.view, d.fr.N);
    if (ln < kEps || vn < kEps) return;
    V3 M = normalize(L + d.view);
    float lm = dot(L, M);
    float nm = dot(d.fr.N, M);
    if (nm < kEps || lm < kEps) return;
    float nm2 = sqr(nm);
    float Ds = D_GTR2Aniso(d, M, nm2);
    float FH = R_POW5(clampf(1.0f - lm, 0.0f, 1.0f));
    float FsR = lerpf(FH, d.f0R, 1.0f);
    float FsG = lerpf(FH, d.f0G, 1.0f);
    float FsB = lerpf(FH, d.f0B, 1.0f);
    float Gs = smithG_GGX(ln, d.specRough) * smithG_GGX(vn, d.specRough);
    float Dr = D_GTR1(d, nm2);
    float Fr = lerpf(FH, 0.04f, 1.0f);
    float Gr = smithG_GGX(ln, 0.25f) * smithG_GGX(vn, 0.25f);
    float om = 1.0f - d.metallic;
    float cc = d.clearcoat * Dr * Fr * Gr;
    r = (Ds * FsR * Gs + cc) + FH * d.shR * om;
    g = (Ds * FsG * Gs + cc) + FH * d.shG * om;
    b = (Ds * FsB * Gs + cc) + FH * d.shB * om;
}

// sampleGTR1Direction, src/rlDisney.cpp:393-404
RLS_DEV V3 disney_gtr1_microfacet(const Disney &d, float rx, float ry)
{
    float phiH = kTwoPi * rx;
    float a2 = sqr(d.roughness);
    float cosThetaH = a2 == 1.0f
        ? R_SQRT1M(ry)
        : R_SQRT(R_DIV(1.0f - R_POW(a2, 1.0f - ry), 1.0f - a2));
    V3 omega = spherical_direction(cosThetaH, phiH);
    return normalize(to_frame(omega, d.fr.U, d.fr.V, d.fr.N));
}

// Alternates the reference compiles but never selects (mSampleFromVisibleNormal = true, src/rlDisney.cpp:191)
// sampleGTR2AnisoDirection, src/rlDisney.cpp:406-414
RLS_DEV V3 disney_gtr2_aniso_microfacet(const Disney &d, float rx, float ry)
{
    float gg = R_SQRT(R_DIV(ry, 1.0f - ry));
    float phi = kTwoPi * rx;
    float s, c;
    t_sincos_any(phi, &s, &c);
    V3 omega = mk(gg * d.ax * c, gg * d.ay * s, 1.0f);
    return normalize(to_frame(omega, d.fr.U, d.fr.V, d.fr.N));
}
// sampleGTR2Direction, src/rlDisney.cpp:504-512 (returns the rotated direction without normalising)
RLS_DEV V3 disney_gtr2_direction(const Disney &d, float rx, float ry)
{
    float t = d.roughness * R_SQRT(R_DIV(rx, 1.0f - rx));
#if RLS_FAST
    float cosTheta = __builtin_amdgcn_rsqf(1.0f + t * t);              // cos(atan(t))
#else
    float st, cosTheta;
    t_sincos(rlm::atan32_v(t), &st, &cosTheta);                         // cosf(atanf(t))
#endif
    V3 omega = spherical_direction(cosTheta, kTwoPi * ry);
    return to_frame(omega, d.fr.U, d.fr.V, d.fr.N);
}
// D_GTR2, src/rlDisney.cpp:553-559
RLS_DEV float D_GTR2(const Disney &d, V3 m)
{
    float mn = dot(m, d.fr.N);
    float a2 = sqr(d.roughness);
    float den = kPi * sqr(1.0f + (a2 - 1.0f) * sqr(mn));
    return R_DIV(a2, den);
}
// evalSpecularPdf with mSampleFromVisibleNormal == false, src/rlDisney.cpp:520-532,541-542
RLS_DEV float disney_specular_pdf_ndf(const Disney &d, V3 i)
{
    V3 m = normalize(i + d.view);
    float im = absf(dot(i, m));
    float mn = dot(m, d.fr.N);
    if (mn < 0.0f) return 0.0f;
    float mn2 = sqr(mn);
    float ccw = R_DIV(d.clearcoat, d.clearcoat + 1.0f);
    float D = lerpf(ccw, D_GTR2Aniso(d, m, mn2), D_GTR1(d, mn2));
    return R_DIV(D * absf(mn) * 0.25f, im);
}

// GaussianProfile, src/rlSss.h:63-97 (not instantiated by the reference; Arnold's closed fast_exp -> exp)
struct GaussProfile { float variance, maxR, norm; };
RLS_DEV GaussProfile gauss_make(float dist_x)
{
    GaussProfile g;
    g.maxR = dist_x;
    g.variance = R_DIV(sqr(g.maxR), 12.46f);
    g.norm = 1.0f - R_EXP(R_DIV(-sqr(g.maxR) * 0.5f, g.variance));
    return g;
}
RLS_DEV float gauss_radius(const GaussProfile &g, float rx) { return R_SQRT(-2.0f * g.variance * R_LOG(1.0f - rx * g.norm)); }
RLS_DEV float gauss_profile(const GaussProfile &g, float r)
{
    return R_DIV(0.15915494f, g.variance) * R_EXP(R_DIV(-r * r * 0.5f, g.variance));
}
RLS_DEV float gauss_pdf(const GaussProfile &g, float r) { return R_DIV(gauss_profile(g, r), g.norm); }

// sampleSpecularDirection, src/rlDisney.cpp:367-390 (visible-normal branch; 191: always true)
RLS_DEV V3 disney_sample_specular(const Disney &d, const VndfView &w, float rx, float ry)
{
    V3 M;
    float gtr2Weight = d.gtr2Weight;                 // disney_prepare()
    // rx / w for the GTR2 lobe, (rx - w) / (1 - w) for the clearcoat lobe: one division of selected operands
    const bool gtr2 = rx < gtr2Weight;
    rx = R_DIV(gtr2 ? rx : rx - gtr2Weight, gtr2 ? gtr2Weight : 1.0f - gtr2Weight);
    if (gtr2) {
        M = vndf_microfacet(w, d.fr, rx, ry);
    } else {
        M = disney_gtr1_microfacet(d, rx, ry);
    }
    if (dot(d.fr.N, M) < 0.0f) return mk(0.0f, 0.0f, 0.0f);
    return reflect_direction(d.view, M);
}

// ---- rare branches of the per-sample samplers, evaluated packed ---------------------------------------------------
// Two branches of the samplers are taken by a few lanes of a wavefront and paid for by all 64 at every sample of an
// n^2-spp loop: the uniform-slope fallback of visible-normal sampling (~7 % of the lanes of a mixed workload) and
// rlDisney's clearcoat half vector (sampleGTR1Direction: the lanes whose random number falls beyond gtr2Weight).  Both
// are "a radius from one random number, an azimuth 2 pi x the other": an exact division, one or two exact square roots,
// an fp64 sincosf -- and powf for the clearcoat lobe.  slow_eval is that common form; the n^2-spp loops queue the
// requests of K samples per wavefront in LDS and evaluate them 64 at a time (rls_loops.hpp, SlowLds): the same functions
// on the same arguments, on another lane.  Measured bound (the branches replaced by nothing,
// profiles/r02_valu_rates.txt): rlDisney 64 spp -15 % (clearcoat) and -6 % (fallback); rlGgx loops -5 %.
struct SlowOut { float x, y, z; };
// azimuth number p, radius number q; t = a2 >= 0: clearcoat half vector, t < 0: uniform slope

RLS_DEV SlowOut slow_eval(float p, float q, float t)
{
    SlowOut o;
    float s, c;
    t_sincos_any(kTwoPi * p, &s, &c);
    float r;
    if (t < 0.0f) {                                  // uniform_slope(rx = q, ry = p)
        r = R_SQRT(ratio_to_one_minus(q));
        o.z = 0.0f;
    } else {                                         // disney_gtr1_microfacet's omega (a2 = t, rx = p, ry = q)
        const float cosThetaH = t == 1.0f ? R_SQRT1M(q) : R_SQRT(R_DIV(1.0f - R_POW(t, 1.0f - q), 1.0f - t));
        r = R_SQRT1M(sqr(cosThetaH));
        o.z = cosThetaH;
    }
    o.x = r * c;
    o.y = r * s;
    return o;
}

// evalDiffusePdf, src/rlDisney.cpp:515-518
RLS_DEV float disney_diffuse_pdf(const Disney &d, V3 i) { return maxf(1e-4f, dot(i, d.fr.N) * kInvPi); }

// evalSpecularPdf, src/rlDisney.cpp:520-543 (visible-normal branch)
RLS_DEV float disney_specular_pdf(const Disney &d, V3 i)
{
    V3 m = normalize(i + d.view);
    float im = absf(dot(i, m));
    float mn = dot(m, d.fr.N);
    if (mn < 0.0f) return 0.0f;
    float mn2 = sqr(mn);
    float ccw = R_DIV(d.clearcoat, d.clearcoat + 1.0f);
    float vn = maxf(1e-4f, dot(d.view, d.fr.N));
    float Dw = R_DIV(smithG_GGX(im, d.specRough) * D_GTR2Aniso(d, m, mn2) * 2.0f * im, vn);
    float D = lerpf(ccw, Dw, R_DIV(D_GTR1(d, mn2) * absf(mn), im));
    return D * 0.25f;
}

// static triple dispatch, src/rlDisney.cpp:120-152
template <bool DIFFUSE>
RLS_DEV void disney_eval(const Disney &d, V3 L, float &r, float &g, float &b)
{
    if (is_zero(L)) { r = 0.0f; g = 0.0f; b = 0.0f; return; }
    float nl = dot(d.fr.N, L);
    if (DIFFUSE) disney_eval_diffuse(d, L, r, g, b);
    else disney_eval_specular(d, L, r, g, b);
    r *= nl; g *= nl; b *= nl;
}
template <bool DIFFUSE>
RLS_DEV float disney_pdf(const Disney &d, V3 L)
{
    if (is_zero(L)) return 0.0f;
    return DIFFUSE ? disney_diffuse_pdf(d, L) : disney_specular_pdf(d, L);
}

// evalBrdf and evalPdf of one direction in one go, on a prepared closure (disney_prepare): what the two verbs share
// -- the half vector normalize(L + V), D_GTR2Aniso and D_GTR1 of it -- is computed once, what depends on the closure
// only comes from the Disney struct.  Every value is the one the separate verbs compute.
template <bool DIFFUSE, bool WANT_F, bool WANT_PDF>
RLS_DEV void disney_eval_pdf(const Disney &d, V3 L, float &r, float &g, float &b, float &pdf)
{
    r = 0.0f; g = 0.0f; b = 0.0f; pdf = 0.0f;
    if (is_zero(L)) return;                                          // src/rlDisney.cpp:124-127,141-144
    const float ln = dot(L, d.fr.N);                                 // == dot(N, L) of evalBrdf (136)
    const float vn = d.vn;
    // evalBrdf multiplies whatever evalDiffuse / evalSpecular return by N.L (136): their early "black" comes out as
    // 0 * N.L -- -0 below the horizon, NaN for a NaN direction -- and so it does here
    if (WANT_F) { r = 0.0f * ln; g = r; b = r; }
    if (DIFFUSE) {
        if (WANT_PDF) pdf = maxf(1e-4f, ln * kInvPi);                // evalDiffusePdf, 515-518
        if (!WANT_F || ln < kEps || vn < kEps) return;
        V3 H = normalize(L + d.view);
        float lh = dot(L, H);
        float vh = dot(d.view, H);
        if (vh < kEps || lh < kEps) return;
        float lh2 = sqr(lh);
        float FL = R_POW5(clampf(1.0f - ln, 0.0f, 1.0f));
        float FV = d.FV;
        float F90 = 0.5f + 2.0f * d.roughness * lh2;
        float diffuseFactor = lerpf(FL, 1.0f, F90) * lerpf(FV, 1.0f, F90);
        float Fss90 = d.roughness * lh2;
        float Fss = lerpf(FL, 1.0f, Fss90) * lerpf(FV, 1.0f, Fss90);
        float ssFactor = 1.25f * (Fss * (R_RCP(ln + vn) - 0.5f) + 0.5f);
        float mix = lerpf(d.subsurface, diffuseFactor, ssFactor);
        r = d.baseR * kInvPi * mix * d.om * ln;
        g = d.baseG * kInvPi * mix * d.om * ln;
        b = d.baseB * kInvPi * mix * d.om * ln;
        return;
    }
    V3 M = normalize(L + d.view);                                    // 325 and 527: the same vector
    const float lm = dot(L, M);
    const float nm = dot(d.fr.N, M);
    const float nm2 = sqr(nm);
    const float Ds = D_GTR2Aniso(d, M, nm2);
    const float Dr = D_GTR1_prepared(d, nm2);
    if (WANT_PDF && !(nm < 0.0f)) {                                  // evalSpecularPdf, 520-543
        float im = absf(lm);
        float Dw = R_DIV(smithG_GGX(im, d.specRough) * Ds * 2.0f * im, d.vnc);
        float D = lerpf(d.ccw, Dw, R_DIV(Dr * absf(nm), im));
        pdf = D * 0.25f;
    }
    if (WANT_F) {                                                    // evalSpecular, 318-356
        if (ln < kEps || vn < kEps) return;
        if (nm < kEps || lm < kEps) return;
        float FH = R_POW5(clampf(1.0f - lm, 0.0f, 1.0f));
        float FsR = lerpf(FH, d.f0R, 1.0f);
        float FsG = lerpf(FH, d.f0G, 1.0f);
        float FsB = lerpf(FH, d.f0B, 1.0f);
        float Gs = smithG_GGX(ln, d.specRough) * d.gsV;
        float Fr = lerpf(FH, 0.04f, 1.0f);
        float Gr = smithG_GGX(ln, 0.25f) * d.grV;
        float cc = d.clearcoat * Dr * Fr * Gr;
        r = ((Ds * FsR * Gs + cc) + FH * d.shR * d.om) * ln;
        g = ((Ds * FsG * Gs + cc) + FH * d.shG * d.om) * ln;
        b = ((Ds * FsB * Gs + cc) + FH * d.shB * d.om) * ln;
    }
}

// ---- rlSss: NDProfile, src/rlSss.h:27-61, src/rlSss.cpp:20-106 ---------------------------------
struct NdProfile {
    float d[3], c1[3], c2[3];
    float maxR;
#if !RLS_FAST
    // getPdf divides by max(d_i, AI_EPSILON) twice and by c1_i + 3 c2_i once per channel, at every call: their correctly
    // rounded reciprocals, and whether they sit in the window of rlm::div32_y (2^-14 .. 2^14).  window: 0 none kept,
    // 1 the three of d_i (one-sample kernels: setDistance's -maxR / d_i and getPdf's two divisions by d_i share them),
    // 2 those of c1_i + 3 c2_i as well (the probe-ray loops, which call getPdf three times per hit)
    float dm[3], ydm[3], cw[3], ycw[3];
    int window;
#endif
};

// RECIPROCALS: the profile is evaluated many times (the probe-ray loop of integrateScatter): keep the reciprocals of all of
// getPdf's per-point denominators.  One-sample kernels keep the three of d_i only: each serves three divisions there
// (setDistance's -maxR / d_i, getPdf's -r / d_i and (e1 + e2) / d_i) -- three reciprocals (4 instructions each) and nine
// five-instruction quotients instead of nine IEEE divisions; the three of c1 + 3 c2 would serve one division each.
// expf's range tests ONCE for the several calls NDProfile makes on related arguments (rls_libm.hpp, exp32_in_range_3) instead
// of a compare and a branch per call.  Measured per kernel, each against the same sources without it (tools/ab.sh, one box,
// profiles/r03_exp_range_once.txt): in getPdf as the probe-ray loops call it (nd_pdf) integrateScatter -10 %, rlSkin's
// shader_evaluate -2 %; in setDistance (nd_make) the rlSss probe -2 %, NDProfile alone -2 %, but rlSkin's one-sample kernel
// +5 % at the occupancy the compiler picks (it sits on a register-allocation edge: skin.hip pins the occupancy instead and
// keeps the tests, -2 %); in getPdf + evalProfile of the one-sample kernels
// (nd_pdf_profile_t), once the divisions by 3 had lost their own tests (R_DIVCW), the rlSss probe -2 %, with a uniform scatter
// distance -3.5 %, NDProfile alone -4 %, rlSkin's one-sample kernel +4.5 % unless its occupancy is pinned (skin.hip).  evalProfile alone, which the
// probe-ray loops call per shaded hit (nd_profile: the reciprocals' quotients + one range test): integrateScatter -6 % more,
// rlSkin's shader_evaluate -8 % more.
template <bool RECIPROCALS = false>
RLS_DEV NdProfile nd_make(float dx, float dy, float dz)
{
    NdProfile p;
    p.d[0] = dx; p.d[1] = dy; p.d[2] = dz;
    p.maxR = maxf(dx, maxf(dy, dz)) * 3.0f;
#if !RLS_FAST
    // d_i itself (not max(d_i, AI_EPSILON)) is what setDistance divides by: inside the window the two are the same value
    p.window = 1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        p.dm[i] = maxf(p.d[i], kEps);
        p.ydm[i] = R_RCPW(p.dm[i]);                  // windowed below; outside it the reciprocals are not used
        if (!(p.d[i] >= 0x1p-13f && p.d[i] <= 0x1p14f)) p.window = 0;      // 2^-13 > AI_EPSILON: d_i == max(d_i, AI_EPSILON)
    }
    // -maxR / d_i: maxR = 3 max(d) is in [3 x 2^-13, 3 x 2^14] whenever the window holds (or NaN, which fails it).  expf's
    // range tests once for the six calls: |q / 3| <= |q|, so the three quotients bound all six arguments
    float q[3] = { 0.0f, 0.0f, 0.0f };
    if (p.window != 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = rlm::div32_y(-p.maxR, p.d[i], p.ydm[i]);
    }
    if (__builtin_expect(p.window != 0 && rlm::exp32_in_range_3(q[0], q[1], q[2]), 1)) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            p.c1[i] = 1.0f - R_EXP_IN_RANGE(q[i]);
            // (|q| is between 3 x 2^-13 / 2^14 and 3 x 2^14 / 2^-13 inside the window: the division by 3 needs no range test)
            p.c2[i] = 1.0f - R_EXP_IN_RANGE(R_DIVCW(q[i], 3.0f));
        }
    } else
#endif
    {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            p.c1[i] = 1.0f - R_EXP(R_DIV(-p.maxR, p.d[i]));
            p.c2[i] = 1.0f - R_EXP(R_DIVC(R_DIV(-p.maxR, p.d[i]), 3.0f));
        }
    }
#if !RLS_FAST
#pragma unroll
    for (int i = 0; RECIPROCALS && i < 3; i++) {
        p.cw[i] = p.c1[i] + p.c2[i] * 3.0f;
        p.ycw[i] = R_RCPW(p.cw[i]);
        if (!(p.cw[i] >= 0x1p-14f && p.cw[i] <= 0x1p14f)) p.window = 0;
    }
    if (RECIPROCALS && p.window) p.window = 2;
#endif
    return p;
}

RLS_DEV NdProfile nd_wave_uniform(NdProfile p)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        p.d[i] = wave_uniform(p.d[i]); p.c1[i] = wave_uniform(p.c1[i]); p.c2[i] = wave_uniform(p.c2[i]);
#if !RLS_FAST
        p.dm[i] = wave_uniform(p.dm[i]); p.ydm[i] = wave_uniform(p.ydm[i]);
        p.cw[i] = wave_uniform(p.cw[i]); p.ycw[i] = wave_uniform(p.ycw[i]);
#endif
    }
    p.maxR = wave_uniform(p.maxR);
#if !RLS_FAST
    p.window = wave_uniform(p.window);
#endif
    return p;
}

// Tried and not kept (the code is in history at 4127ff2, switches RLS_ND_MERGE_RADIUS / RLS_ND_RADIUS_SELECTS): one division
// and one logf of selected operands for getRadius's two arms -- measured twice, rounds 2 and 3: the rlSss probe 1.576 -> 1.643
// ms, rlSkin 4.12 -> 4.17, slower although it executes fewer instructions; the channel lottery by selects: the probe +27 %.
// selectDistLobe + getRadius, src/rlSss.h:30-42, src/rlSss.cpp:36-66
RLS_DEV float nd_radius(const NdProfile &p, float rx)
{
    if (p.maxR < kEps) return 0.0f;
    float d, w1, w2;
    // LINEARSTEP(lo, hi, t) = CLAMP((t - lo) / (hi - lo), 0, 1) with constant bounds: division by a constant
    if (rx < 0.3333f) {
        rx = clampf(R_DIVC(rx - 0.0f, 0.3333f - 0.0f), 0.0f, 1.0f);
        d = p.d[0]; w1 = p.c1[0]; w2 = p.c2[0];
    } else if (rx > 0.6666f) {
        rx = clampf(R_DIVC(rx - 0.6666f, 1.0f - 0.6666f), 0.0f, 1.0f);
        d = p.d[2]; w1 = p.c1[2]; w2 = p.c2[2];
    } else {
        rx = clampf(R_DIVC(rx - 0.3333f, 0.6666f - 0.3333f), 0.0f, 1.0f);
        d = p.d[1]; w1 = p.c1[1]; w2 = p.c2[1];
    }
    if (d < kEps) return 0.0f;
    float w = R_DIV(w1, w1 + w2 * 3.0f);
    float r;
    if (rx > w) {
        rx = linearstep(w, 1.0f, rx);
        r = R_LOG(1.0f - rx * w2) * (-d * 3.0f);
    } else {
        rx = linearstep(0.0f, w, rx);
        r = R_LOG(1.0f - rx * w1) * (-d);
    }
    return r;
}

// getPdf, src/rlSss.cpp:68-84
RLS_DEV float nd_pdf_ieee(const NdProfile &p, float r)
{
    float pdf = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float d = maxf(p.d[i], kEps);
        float p1 = R_EXP(R_DIV(-r, d));
        float p2 = R_EXP(R_DIVC(R_DIV(-r, d), 3.0f));
        pdf += R_DIV(R_DIV(p1 + p2, d), p.c1[i] + p.c2[i] * 3.0f);
    }
    return R_DIV(pdf, kTwoPi * r * 3.0f);
}
RLS_DEV float nd_pdf(const NdProfile &p, float r)
{
    if (p.maxR < kEps) return 1.0f;
#if RLS_FAST
    return nd_pdf_ieee(p, r);
#else
    // Nine of the ten divisions have a per-point denominator: through its reciprocal (rlm::div32_y, five instructions
    // each) when every operand is inside that routine's window -- the point's denominators (p.window), r, and the sums
    // e^(-r/d) + e^(-r/3d), which fall below 2^-60 only for r > 41 d; the quotients (p1 + p2) / d are then >= 2^-74
    if (__builtin_expect(!(p.window == 2 && r >= 0x1p-40f && r <= 0x1p40f), 0)) return nd_pdf_ieee(p, r);
    float s[3], q[3];
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = rlm::div32_y(-r, p.dm[i], p.ydm[i]);
    if (__builtin_expect(!rlm::exp32_in_range_3(q[0], q[1], q[2]), 0)) return nd_pdf_ieee(p, r);     // expf's range tests, once
#pragma unroll
    for (int i = 0; i < 3; i++) s[i] = R_EXP_IN_RANGE(q[i]) + R_EXP_IN_RANGE(R_DIVCW(q[i], 3.0f));      // 2^-54 <= |q| < 88
    if (__builtin_expect(!(minf(s[0], minf(s[1], s[2])) >= 0x1p-60f), 0)) return nd_pdf_ieee(p, r);
    float pdf = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) pdf += rlm::div32_y(rlm::div32_y(s[i], p.dm[i], p.ydm[i]), p.cw[i], p.ycw[i]);
    return R_DIV(pdf, kTwoPi * r * 3.0f);
#endif
}

// getPdf at three radii (the 3-axis MIS pdf of a probe hit asks for them together): nd_pdf()'s tests -- the window, the radii's
// range, expf's range, the sums' floor -- once for the three calls instead of once each; the same operations per radius and
// channel, so the same bits, and any failing test sends all three through nd_pdf()
RLS_DEV void nd_pdf3(const NdProfile &p, float r0, float r1, float r2, float (&out)[3])
{
    out[0] = nd_pdf(p, r0); out[1] = nd_pdf(p, r1); out[2] = nd_pdf(p, r2);
}

// evalProfile, src/rlSss.cpp:86-106
RLS_DEV void nd_profile(const NdProfile &p, float r, float &R, float &G, float &B)
{
    if (p.maxR < kEps) { R = 0.0f; G = 0.0f; B = 0.0f; return; }
    if (r < kEps) { R = 1.0f; G = 1.0f; B = 1.0f; return; }
    float denom = 8.0f * kPi * r;
    float out[3];
#if !RLS_FAST
    // inside the window of the per-point reciprocals (d_i >= 2^-13 > AI_EPSILON: the d_i < AI_EPSILON arm cannot occur) the three
    // quotients -r / d_i through them, and expf's range tests once for the six calls (|-r / (3 d_i)| <= |-r / d_i|)
    if (__builtin_expect(p.window != 0 && r <= 0x1p40f, 1)) {
        float q[3];
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = rlm::div32_y(-r, p.dm[i], p.ydm[i]);
        if (__builtin_expect(rlm::exp32_in_range_3(q[0], q[1], q[2]), 1)) {
#pragma unroll
            for (int i = 0; i < 3; i++)
                out[i] = R_DIV(R_EXP_IN_RANGE(q[i]) + R_EXP_IN_RANGE(R_DIV(-r, 3.0f * p.d[i])), denom * p.d[i]);
            R = out[0]; G = out[1]; B = out[2];
            return;
        }
    }
#endif
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float d = p.d[i];
        out[i] = d < kEps ? 1.0f : R_DIV(R_EXP(R_DIV(-r, d)) + R_EXP(R_DIV(-r, 3.0f * d)), denom * d);
    }
    R = out[0]; G = out[1]; B = out[2];
}

// getPdf and evalProfile at the same radius (the probe-ray sample of rlSss / rlSkin asks for both): e^(-r / d_i) is one
// value in both -- getPdf divides by max(d_i, AI_EPSILON), evalProfile by d_i and only when d_i >= AI_EPSILON -- so
// the three divisions and exponentials are done once.  Same results as nd_pdf() and nd_profile().
// WINDOWED: 0 every division the IEEE way, 1 those by d_i through its reciprocal, 2 those by c1_i + 3 c2_i as well
template <int WINDOWED>
RLS_DEV void nd_pdf_profile_t(const NdProfile &p, float r, float &pdf, float &R, float &G, float &B)
{
    const float denom = 8.0f * kPi * r;
    float acc = 0.0f;
    float out[3];
    bool tiny = false;                                // WINDOWED: a sum e^(-r/d) + e^(-r/3d) below div32_y's window
#if !RLS_FAST
    // ONCE: expf's range tests once for the nine calls -- |q / 3| <= |q| and, d_i being its own max(d_i, AI_EPSILON) inside
    // the window, |-r / (3 d_i)| <= |q| as well: the three quotients bound all nine arguments.  Outside the range: the general form
    constexpr bool ONCE = WINDOWED != 0;
    float qw[3] = { 0.0f, 0.0f, 0.0f };
    if (ONCE) {
#pragma unroll
        for (int i = 0; i < 3; i++) qw[i] = rlm::div32_y(-r, p.dm[i], p.ydm[i]);
        if (__builtin_expect(!rlm::exp32_in_range_3(qw[0], qw[1], qw[2]), 0)) { nd_pdf_profile_t<0>(p, r, pdf, R, G, B); return; }
    }
#endif
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float d = maxf(p.d[i], kEps);
#if !RLS_FAST
        const float q = ONCE ? qw[i] : WINDOWED ? rlm::div32_y(-r, p.dm[i], p.ydm[i]) : R_DIV(-r, d);
        const float p1 = ONCE ? R_EXP_IN_RANGE(q) : R_EXP(q);
        // (WINDOWED: r >= 2^-40 and d_i <= 2^14, so |q| >= 2^-54, and |q| <= 2^53: the division by 3 needs no range test)
        const float p2 = ONCE ? R_EXP_IN_RANGE(R_DIVCW(q, 3.0f)) : WINDOWED ? R_EXP(R_DIVCW(q, 3.0f)) : R_EXP(R_DIVC(q, 3.0f));
#else
        const float q = R_DIV(-r, d);
        const float p1 = R_EXP(q);
        const float p2 = R_EXP(R_DIVC(q, 3.0f));
#endif
#if !RLS_FAST
        if (WINDOWED == 2) {
            tiny = tiny || !(p1 + p2 >= 0x1p-60f);
            acc += rlm::div32_y(rlm::div32_y(p1 + p2, p.dm[i], p.ydm[i]), p.cw[i], p.ycw[i]);
        } else if (WINDOWED == 1) {
            tiny = tiny || !(p1 + p2 >= 0x1p-60f);
            acc += R_DIV(rlm::div32_y(p1 + p2, p.dm[i], p.ydm[i]), p.c1[i] + p.c2[i] * 3.0f);
        } else
#endif
        acc += R_DIV(R_DIV(p1 + p2, d), p.c1[i] + p.c2[i] * 3.0f);
#if !RLS_FAST
        if (ONCE) out[i] = R_DIV(p1 + R_EXP_IN_RANGE(R_DIV(-r, 3.0f * p.d[i])), denom * p.d[i]);     // (d_i >= 2^-13 > AI_EPSILON)
        else
#endif
        out[i] = p.d[i] < kEps ? 1.0f : R_DIV(p1 + R_EXP(R_DIV(-r, 3.0f * p.d[i])), denom * p.d[i]);
    }
    if (WINDOWED && __builtin_expect(tiny, 0)) { nd_pdf_profile_t<0>(p, r, pdf, R, G, B); return; }
    pdf = R_DIV(acc, kTwoPi * r * 3.0f);
    const bool white = r < kEps;
    R = white ? 1.0f : out[0]; G = white ? 1.0f : out[1]; B = white ? 1.0f : out[2];
}
RLS_DEV void nd_pdf_profile(const NdProfile &p, float r, float &pdf, float &R, float &G, float &B)
{
    if (p.maxR < kEps) { pdf = 1.0f; R = 0.0f; G = 0.0f; B = 0.0f; return; }
#if !RLS_FAST
    // the divisions of getPdf by per-point denominators through their reciprocals, as nd_pdf()
    if (__builtin_expect(p.window != 0 && r >= 0x1p-40f && r <= 0x1p40f, 1)) {
        if (p.window == 2) nd_pdf_profile_t<2>(p, r, pdf, R, G, B);
        else nd_pdf_profile_t<1>(p, r, pdf, R, G, B);
        return;
    }
#endif
    nd_pdf_profile_t<0>(p, r, pdf, R, G, B);
}

// SssSampler frame, src/rlSss.h:149-158
RLS_DEV Frame sss_frame(V3 Ns, V3 t, bool has_dPdu)
{
    Frame fr;
    fr.N = Ns;
    if (has_dPdu && !is_zero(t) && is_finite3(t)) {
        V3 U = normalize(t);
        fr.V = normalize(cross(Ns, U));
        fr.U = cross(fr.V, Ns);
    } else {
        fr.U = t;
        fr.V = cross(Ns, t);
    }
    return fr;
}

// getProbeRay, src/rlSss.h:487-533
// Tried and not kept (history at 4127ff2, switch RLS_PROBE_SELECTS): the axis lottery without branches -- the three linearstep()
// divisors are powers of two and the three to_frame() calls differ only in the axes they read, so one call on selected axes
// gives the same bits; measured: nothing, +-0.3 % on every kernel that draws probe rays.
RLS_DEV float sss_probe_ray(const NdProfile &p, const Frame &fr, float rx, float ry,
                            V3 &offset, V3 &dir, float &maxdist)
{
    int idx;
    if (rx < 0.5f) {
        idx = 0;
        rx = linearstep(0.0f, 0.5f, rx);
    } else if (rx < 0.75f) {
        idx = 2;
        rx = linearstep(0.5f, 0.75f, rx);
    } else {
        idx = 3;
        rx = linearstep(0.75f, 1.0f, rx);
    }
    float r = nd_radius(p, rx);
    float rmax = p.maxR;
    float phi = kTwoPi * ry;
    float s, c;
    t_sincos_any(phi, &s, &c);
    V3 o;
    o.x = c * r;
    o.z = s * r;
    o.y = R_SQRT(rmax * rmax - r * r);
    maxdist = o.y * 2.0f;
    if (idx < 2) {
        dir = -fr.N;
        offset = to_frame(o, fr.U, -dir, fr.V);
    } else if (idx == 2) {
        dir = fr.U;
        offset = to_frame(o, fr.V, -dir, fr.N);
    } else {
        dir = fr.V;
        offset = to_frame(o, fr.N, -dir, fr.U);
    }
    return r;
}

// 3-axis MIS pdf, src/rlSss.h:246-266
RLS_DEV float sss_mis_pdf(const NdProfile &p, const Frame &fr, V3 disp, V3 sN, bool literal)
{
    V3 o = literal ? to_frame(disp, fr.U, fr.V, fr.N)
                   : mk(dot(disp, fr.U), dot(disp, fr.V), dot(disp, fr.N));
    o = mk(o.x * o.x, o.y * o.y, o.z * o.z);
    float rr0 = R_SQRT(o.y + o.z);
    float rr1 = R_SQRT(o.x + o.z);
    float rr2 = R_SQRT(o.x + o.y);
    float pd[3];
    nd_pdf3(p, rr0, rr1, rr2, pd);
    return pd[0] * absf(dot(fr.U, sN)) * 0.25f
         + pd[1] * absf(dot(fr.V, sN)) * 0.25f
         + pd[2] * absf(dot(fr.N, sN)) * 0.5f;
}

// cavity fade, src/rlSss.h:401-413
RLS_DEV float sss_cavity_fade(V3 disp, float r, V3 sN, V3 No)
{
    V3 dd = mk(R_DIV(disp.x, r), R_DIV(disp.y, r), R_DIV(disp.z, r));
    float c = dot(No, dd) < 0.0f ? absf(dot(sN, No)) : clampf(dot(sN, No), -1.0f, 1.0f);
    return R_SQRT((1.0f + c) * 0.5f);
}

// ---- rlGgx direct lighting: stand-ins for the closed light services (include/rlshaders_amd.h,
// rls_ggx_direct_lighting) ------------------------------------------------------------------------
// qualitative Oren-Nayar (SIGGRAPH'94), BRDF x cos(theta_i); cosine-weighted pdf
struct OrenNayar { V3 N; float A, B; };
RLS_DEV OrenNayar oren_nayar_make(V3 N, float sigma)
{
    OrenNayar o;
    float s2 = sigma * sigma;
    o.N = N;
    o.A = 1.0f - 0.5f * R_DIV(s2, s2 + 0.33f);
    o.B = 0.45f * R_DIV(s2, s2 + 0.09f);
    return o;
}
RLS_DEV float oren_nayar_brdf(const OrenNayar &o, V3 wo, V3 wi)
{
    float ci = dot(o.N, wi), co = dot(o.N, wo);
    if (!(ci > 0.0f) || !(co > 0.0f)) return 0.0f;
    float si = R_SQRT(maxf(0.0f, 1.0f - ci * ci)), so = R_SQRT(maxf(0.0f, 1.0f - co * co));
    float cphi = 0.0f;
    if (si > kEps && so > kEps) cphi = maxf(0.0f, R_DIV(dot(wi, wo) - ci * co, si * so));
    float sinAlpha, tanBeta;
    if (ci > co) { sinAlpha = so; tanBeta = R_DIV(si, ci); }
    else         { sinAlpha = si; tanBeta = R_DIV(so, co); }
    return kInvPi * (o.A + o.B * cphi * sinAlpha * tanBeta) * ci;
}
RLS_DEV float oren_nayar_pdf(const OrenNayar &o, V3 wi)
{
    float ci = dot(o.N, wi);
    return ci > 0.0f ? ci * kInvPi : 0.0f;
}

// the cone a spherical light subtends from P: axis w, basis (u, v) (Duff et al. 2017), uniform pdf
struct LightCone { bool valid; V3 d, u, v, w; float c2, cosMax, pdf; };
RLS_DEV LightCone cone_make(V3 center, float radius, V3 P)
{
    LightCone c;
    c.d = center - P;
    float dist2 = dot(c.d, c.d), r2 = radius * radius;
    c.c2 = dist2 - r2;
    c.valid = c.c2 > 0.0f;
    float sin2 = R_DIV(r2, dist2);
    c.cosMax = R_SQRT(maxf(0.0f, 1.0f - sin2));
    c.pdf = R_RCP(kTwoPi * R_DIV(sin2, 1.0f + c.cosMax));
    float inv = R_RCPW(R_SQRT(dist2));            // of a square root
    c.w = c.d * inv;
    float sg = __builtin_copysignf(1.0f, c.w.z);
    float a = R_DIV(-1.0f, sg + c.w.z);
    float b = c.w.x * c.w.y * a;
    c.u = mk(1.0f + sg * c.w.x * c.w.x * a, sg * b, -sg * c.w.x);
    c.v = mk(b, sg + c.w.y * c.w.y * a, -c.w.y);
    return c;
}
RLS_DEV V3 cone_sample(const LightCone &c, float rx, float ry)
{
    float ct = 1.0f - rx * (1.0f - c.cosMax);
    float st = R_SQRT(maxf(0.0f, 1.0f - ct * ct));
    float sn, cs;
    t_sincos(kTwoPi * ry, &sn, &cs);
    float x = st * cs, y = st * sn;
    return c.u * x + c.v * y + c.w * ct;
}
RLS_DEV bool cone_hit(const LightCone &c, V3 dir)
{
    float b = dot(c.d, dir);
    return b > 0.0f && !(b * b - c.c2 * dot(dir, dir) < 0.0f);
}
RLS_DEV float power_heuristic(float pa, float pb) { return R_DIV(pa * pa, pa * pa + pb * pb); }

// ---- counter-based generator: integer hash + exactly rounded ops only (reproducible on a CPU) ----
RLS_DEV uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}
RLS_DEV uint32_t hash_u32(uint32_t seed, uint64_t index, uint32_t stream)
{
    uint32_t h = mix32(seed ^ (0x9E3779B9U * (stream + 1U)));
    h = mix32(h ^ (uint32_t)(index & 0xFFFFFFFFULL));
    h = mix32(h + (uint32_t)(index >> 32) * 0x85EBCA6BU + 0xC2B2AE35U);
    return h;
}
RLS_DEV float hash_u01(uint32_t seed, uint64_t index, uint32_t stream)
{
    return (float)(hash_u32(seed, index, stream) >> 8) * (1.0f / 16777216.0f);
}

// ---- streaming loads / stores -------------------------------------------------------------------
// Every plane is read or written exactly once per launch: non-temporal so the streams do not
// evict each other from L2 / Infinity Cache.
// Cache policy of the plane accesses: every plane is touched exactly once per launch, so both directions are non-temporal.
// Plain loads, plain stores and both were measured in round 5, after the pointwise kernels turned out to run at the board's
// power cap (a policy that spares the caches' energy would come back as clock): profiles/r05_mem_policy.txt -- not kept.
template <class P> RLS_DEV float ld_policy(P p) { return __builtin_nontemporal_load(p); }
template <class P> RLS_DEV void st_policy(float v, P p) { __builtin_nontemporal_store(v, p); }
RLS_DEV float ldg(const float *p, int64_t i) { return ld_policy(p + i); }
RLS_DEV void stg(float *p, int64_t i, float v) { st_policy(v, p + i); }

// A point index split into a wave-uniform 64-bit part and a 32-bit lane part: `p + base` is scalar
// arithmetic and the access becomes global_load_dword v, v_lane_offset, s[base] -- no 64-bit vector
// add per plane (31 planes per point in the reflect+refract kernel).
struct Idx {
    int64_t base;     // first point of this workgroup's tile in this iteration
    uint32_t lane;    // threadIdx.x
    uint32_t byte;    // 4 * threadIdx.x, the 32-bit vector offset of the access
    RLS_DEV int64_t full() const { return base + (int64_t)lane; }
};
// The empty asm keeps the zero-extension of the lane's byte offset inside the loop body: hoisted out
// of it (as a 64-bit register pair) instruction selection no longer sees "uniform base + zext(32-bit
// offset)" and falls back to a 64-bit vector add per access.
RLS_DEV Idx make_idx(int64_t base)
{
    uint32_t byte = threadIdx.x * 4u;
    asm volatile("" : "+v"(byte));
    Idx i = { base, threadIdx.x, byte };
    return i;
}
// plane pointer + tile offset: a wave-uniform sum, formed with scalar arithmetic and KEPT in a scalar register pair (the
// empty asm) -- left alone, the optimiser re-associates it to (pointer + lane offset) + tile offset whenever the pointer
// is loaded inside the loop (reload_args), which costs two 64-bit vector adds and two moves per access instead of none.
// The pointer is cast to the global address space first: through the asm it would otherwise come back as a generic
// pointer and the access as a flat_load / flat_store.
// RLS_LOAD_RENEW: loads that sit in a later basic block than make_idx() (after a per-parameter stream-or-uniform branch,
// after reload_args) renew the barrier on the lane offset like the stores do: one move instead of a 64-bit vector add
#ifndef RLS_LOAD_RENEW
#define RLS_LOAD_RENEW 0
#endif
typedef __attribute__((address_space(1))) char GChar;
typedef __attribute__((address_space(1))) float GFloat;
RLS_DEV const GFloat *at(const float *p, Idx i)
{
    const GChar *q = (const GChar *)(p + i.base);
    asm("" : "+s"(q));
    return (const GFloat *)(q + i.byte);
}
RLS_DEV GFloat *at(float *p, Idx i)
{
    GChar *q = (GChar *)(p + i.base);
    asm("" : "+s"(q));
    return (GFloat *)(q + i.byte);
}
RLS_DEV float ldg(const float *p, Idx i)
{
#if RLS_LOAD_RENEW
    asm volatile("" : "+v"(i.byte));
#endif
    return ld_policy(at(p, i));
}
RLS_DEV void stg(float *p, Idx i, float v)
{
    // stores sit in later basic blocks than make_idx(): renew the barrier so the zext is local again
    asm volatile("" : "+v"(i.byte));
    st_policy(v, at(p, i));
}

} // namespace rlsd
