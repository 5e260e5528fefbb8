// rls_loops.hpp -- the n^2-samples-per-point loops shared by integrate.hip, lights.hip, scatter.hip and shade.hip: the in-kernel
// sampler (scrambled (0,2)-sequence table in LDS), sums in sample order over G-lane groups (fold), the LDS queues that pack
// the samplers' rare branches 64 at a time (SlowLds), and per shading point: integrateGlossy (rlGgx, rlDisney both lobes),
// integrateRefract, the light loops of rlGgx and rlDisney, SssSampler::integrateScatter over an analytic scene.  Device
// templates only (everything __forceinline__); the kernels and the C ABI are in the four units.  Citations per function.
#pragma once
#include <stdlib.h>

#include "rls_internal.hpp"

using namespace rlsd;

namespace {

// Occupancy of the integrator kernels.  Left alone the register allocator takes 160-172 VGPRs (3 or 2 waves per SIMD,
// changing with unrelated edits); these loops are chains of dependent arithmetic with LDS table reads in between, and
// four waves hide that better than the extra registers help: measured 3 -> 4 waves: rlDisney 64 spp 102.7 -> 93.3 ms,
// rlSkin shader_evaluate 107.3 -> 95.9 ms, the rlGgx light loop 25.6 -> 25.2 ms; 5, 6 and 8 are slower (spills).
// plane pointers re-read per point (reload_args) in the rlDisney n^2-spp kernel: 126 -> 24 spilled scalar registers, and no
// time (73.06 / 72.90 ms with, 73.10 / 73.13 without): the spills sat outside the sample loop already.  Kept for the registers;
// the other loop kernels were left alone.
#define RLS_INT_ARGS(a) reload_args(a)
#ifndef RLS_INT_WAVES
#define RLS_INT_WAVES 4
#endif
#define RLS_INT_ATTR __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_INT_WAVES, RLS_INT_WAVES)))
// rlDisney's light loop and whole node evaluate packed requests with a second 45-word closure in registers: at four
// waves (128 VGPRs) they spill 123 of them and the packing gains 3 %; at three it gains 17 % / 9 % (60.3 -> 52.1 ms,
// 81.6 -> 73.7 ms)
#ifndef RLS_DISNEY_LIGHT_WAVES
#define RLS_DISNEY_LIGHT_WAVES 3
#endif
#define RLS_DISNEY_LIGHT_ATTR __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_DISNEY_LIGHT_WAVES, RLS_DISNEY_LIGHT_WAVES)))

constexpr int kMaxSpp = 256;   // spp_n <= 16

// samples per pass of the loops that pack their samplers' rare branches (SlowLds below)
#ifndef RLS_SPEC_BLOCK
#define RLS_SPEC_BLOCK 4
#endif

// hash stream ids of the per-point scrambles (DESIGN.md "Synthetic inputs": streams 64..67)
constexpr uint32_t kScrambleStream = 64;

__device__ __forceinline__ uint32_t sobol2(uint32_t s)
{
    uint32_t r = 0;
    for (uint32_t v = 1u << 31; s != 0; s >>= 1, v ^= v >> 1) {
        if (s & 1u) r ^= v;
    }
    return r;
}

__device__ __forceinline__ float bits_u01(uint32_t b) { return (float)(b >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void stage_table(uint32_t (*tab)[kMaxSpp], int spp)
{
    for (int t = threadIdx.x; t < spp; t += rlsh::kBlock) {
        tab[0][t] = __brev((uint32_t)t);
        tab[1][t] = sobol2((uint32_t)t);
    }
    __syncthreads();
}

RLS_DEV V3 arr3(const float (&a)[3]) { return mk(a[0], a[1], a[2]); }

// butterfly sum over the G lanes of a group: for the integer-valued sums only (sample counts), where the order of the
// additions cannot change the result
template <int G>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sums in SAMPLE order whatever the group width.  Lane `sub` of a G-lane group takes samples sub, sub + G, ...: within one
// round of the sample loop the G lanes hold the terms of G consecutive samples.  `acc` is kept replicated in all lanes of
// the group; fold adds the round's terms to it in lane order -- sample order -- so that G = 4, 16, 64 produce, bit for bit,
// the sum the one-lane-per-point loop (and the reference's `result +=` loop) produces.  Every lane of the group calls it at
// the same point of the round, with +0 where it has no term (a sum that starts at +0 is never -0, so adding +0 changes
// nothing); the lanes of other groups of the wavefront may be masked off (rlSkin's per-point branches).  G = 1: a plain add.
// Cost: G cross-lane reads + adds per accumulator and round; G > 1 only runs on batches too small to fill the GPU.
template <int G>
__device__ __forceinline__ float group_lane(float v, int l)      // the value lane l of this lane's group holds
{
    if (G == 64) return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
    return __shfl(v, (int)((threadIdx.x & 63u) & ~(unsigned)(G - 1)) + l, 64);
}
template <int G>
__device__ __forceinline__ void fold(float &acc, float t)
{
    if (G == 1) { acc += t; return; }
#pragma unroll
    for (int l = 0; l < G; l++) acc += group_lane<G>(t, l);
}
// two terms per sample, added as the one-lane loop adds them: sample by sample, t1 then t2
template <int G>
__device__ __forceinline__ void fold2(float &acc, float t1, float t2)
{
    if (G == 1) { acc += t1; acc += t2; return; }
#pragma unroll
    for (int l = 0; l < G; l++) { acc += group_lane<G>(t1, l); acc += group_lane<G>(t2, l); }
}

// ---------------------------------------------------------------------------------------------
// Packed evaluation of the samplers' rare branches through LDS (rls_device.hpp, slow_eval, says which and why): the loop
// takes K samples per pass; in a first sweep every lane runs the common part of each sample and queues what needs the
// rare branch (per wavefront, in LDS), the queue is evaluated 64 requests at a time, and a second sweep picks the
// results up and finishes the samples in order.  Both sweeps are rolled loops -- the K samples' state lives in LDS, not
// in registers -- so the code and the register count stay those of the plain loop.
template <int K>
struct SlowLds {
    float q[rlsh::kBlock / 64][4][K * 64];      // per wavefront: requests (p, q, t[, lane]), overwritten by the results
    float st[4][K][rlsh::kBlock];               // per lane and sample: two values of the caller's + flags | slot << 2;
                                                // [3]: the flags | slot of an evaluation request ([0..2] stay the caller's)
};

__device__ __forceinline__ void wave_lds_fence()   // LDS traffic between the lanes of ONE wavefront: order it, no s_barrier
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// first sweep, sample k: every lane of the wavefront calls this (ballot); flags: the caller's two low bits
template <int K>
__device__ __forceinline__ void slow_push(SlowLds<K> &L, int k, int &cnt, bool want, float p, float q, float t,
                                          float u, float v, int flags)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    const uint64_t m = __builtin_amdgcn_ballot_w64(want);
    const int slot = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (want) { L.q[wave][0][slot] = p; L.q[wave][1][slot] = q; L.q[wave][2][slot] = t; }
    L.st[0][k][tid] = u; L.st[1][k][tid] = v;
    L.st[2][k][tid] = __int_as_float((flags & 1) | (want ? 2 : 0) | (slot << 2));
    cnt += __builtin_popcountll(m);
}
// the queue is worked off by the lanes that are active here (rlSkin runs its lobes inside per-point branches): the
// r-th active lane takes requests r, r + A, ... of the A active lanes
template <int K>
__device__ __forceinline__ void slow_run(SlowLds<K> &L, int cnt)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    const uint64_t ex = __builtin_amdgcn_ballot_w64(true);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u));
    const int nact = __builtin_popcountll(ex);
    wave_lds_fence();
    for (int base = 0; base < cnt; base += nact) {
        const int j = base + rank;
        if (j < cnt) {
            const SlowOut o = slow_eval(L.q[wave][0][j], L.q[wave][1][j], L.q[wave][2][j]);
            L.q[wave][0][j] = o.x; L.q[wave][1][j] = o.y; L.q[wave][2][j] = o.z;
        }
    }
    wave_lds_fence();
}
// second sweep, sample k: the caller's two values, its flag, whether a result was asked for, and the result
template <int K>
__device__ __forceinline__ bool slow_pop(const SlowLds<K> &L, int k, float &u, float &v, int &flag, SlowOut &o)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    u = L.st[0][k][tid]; v = L.st[1][k][tid];
    const int f = __float_as_int(L.st[2][k][tid]);
    flag = f & 1;
    const bool want = (f & 2) != 0;
    o.x = 0.0f; o.y = 0.0f; o.z = 0.0f;
    if (want) { const int slot = f >> 2; o.x = L.q[wave][0][slot]; o.y = L.q[wave][1][slot]; o.z = L.q[wave][2][slot]; }
    return want;
}

// sampleSpecularDirection (src/rlDisney.cpp:367-390) in two halves around the packed evaluation.  First half: the lobe
// pick, the closed-form slopes, the request.  Second half: the microfacet normal from whichever source, the reflection.
// Together they return what disney_sample_specular(d, w, rx, ry) returns.
template <int K>
__device__ __forceinline__ void disney_spec_push(SlowLds<K> &L, int k, int &cnt, bool ok, const Disney &d, const VndfView &w,
                                                 float rx, float ry)
{
    const bool gtr2 = rx < d.gtr2Weight;
    const float num = gtr2 ? rx : rx - d.gtr2Weight, den = gtr2 ? d.gtr2Weight : 1.0f - d.gtr2Weight;
    V2 slope;
    float rxp;
    bool needU;
#if !RLS_FAST
    // rx comes from the in-kernel sampler (a multiple of 2^-24 below 1) and the two denominators are per-point values: the
    // rescaled rx and A = 2 rx' / G1 - 1 through their reciprocals (rlm::div32_y) unless some lane has none
    const float y = gtr2 ? d.yW : d.y1mW;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(y == 0.0f || w.yG1 == 0.0f) == 0ull, 1)) {
        rxp = rlm::div32_y(num, den, y);
        needU = vndf_slope_closed<true>(w, rxp, ry, slope);
    } else
#endif
    {
        rxp = R_DIV(num, den);
        needU = vndf_slope_closed(w, rxp, ry, slope);                        // every lane; used where gtr2
    }
    slow_push<K>(L, k, cnt, ok && (!gtr2 || needU), gtr2 ? ry : rxp, gtr2 ? rxp : ry, gtr2 ? -1.0f : sqr(d.roughness),
                 slope.x, slope.y, gtr2 ? 1 : 0);
}
template <int K>
__device__ __forceinline__ V3 disney_spec_pop(const SlowLds<K> &L, int k, const Disney &d, const VndfView &w)
{
    V2 slope;
    int gtr2;
    SlowOut o;
    const bool got = slow_pop<K>(L, k, slope.x, slope.y, gtr2, o);
    if (gtr2 && got) { slope.x = o.x; slope.y = o.y; }
    V3 M;
#if RLS_FAST
    if (gtr2) M = vndf_from_slope(w, d.fr, slope);
    else M = normalize(to_frame(mk(o.x, o.y, o.z), d.fr.U, d.fr.V, d.fr.N));
#else
    // normalize_h is normalize in EXACT arithmetic: one rotation + normalisation for both sources of omega
    V3 omega;
    omega.x = gtr2 ? -(w.cosPhi * slope.x - w.sinPhi * slope.y) * w.ax : o.x;
    omega.y = gtr2 ? -(w.sinPhi * slope.x + w.cosPhi * slope.y) * w.ay : o.y;
    omega.z = gtr2 ? 1.0f : o.z;
    M = normalize(to_frame(omega, d.fr.U, d.fr.V, d.fr.N));
#endif
    return dot(d.fr.N, M) < 0.0f ? mk(0.0f, 0.0f, 0.0f) : reflect_direction(d.view, M);
}

// VNDFKernel::evalSample (src/rlGgx.cpp:63-99) in two halves around the packed evaluation: the closed-form slopes and the
// request for the uniform-slope fallback; then the microfacet normal.  Together: vndf_microfacet(w, fr, rx, ry).
template <int K>
__device__ __forceinline__ void ggx_vndf_push(SlowLds<K> &L, int k, int &cnt, bool ok, const VndfView &w, float rx, float ry)
{
    V2 slope;
    bool needU;
#if !RLS_FAST
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(w.yG1 == 0.0f) == 0ull, 1))
        needU = vndf_slope_closed<true>(w, rx, ry, slope);       // rx from the in-kernel sampler: G1's reciprocal serves
    else
#endif
        needU = vndf_slope_closed(w, rx, ry, slope);
    slow_push<K>(L, k, cnt, ok && needU, ry, rx, -1.0f, slope.x, slope.y, 1);
}
template <int K>
__device__ __forceinline__ V3 ggx_vndf_pop(const SlowLds<K> &L, int k, const VndfView &w, const Frame &fr)
{
    V2 slope;
    int flag;
    SlowOut o;
    if (slow_pop<K>(L, k, slope.x, slope.y, flag, o)) { slope.x = o.x; slope.y = o.y; }
    return vndf_from_slope(w, fr, slope);
}

// ---------------------------------------------------------------------------------------------
// Evaluation requests: "evalBrdf / evalPdf of MY closure in THIS direction".  The light-sampling strategy of a light
// loop evaluates only the samples above the horizon (half of the lanes of the bench's batches), the BSDF-sampling one
// only those that hit the light (a few per cent), but a wavefront runs the evaluation whenever one lane needs it.
// Queued like the samplers' rare branches (direction + requesting lane), evaluated 64 at a time by lanes that fetch the
// requester's closure across the wavefront (ds_bpermute), results handed back through the queue.
template <int K>
__device__ __forceinline__ void eval_push(SlowLds<K> &L, int k, int &cnt, bool want, V3 dir)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    const uint64_t m = __builtin_amdgcn_ballot_w64(want);
    const int slot = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (want) {
        L.q[wave][0][slot] = dir.x; L.q[wave][1][slot] = dir.y; L.q[wave][2][slot] = dir.z;
        L.q[wave][3][slot] = __int_as_float((tid & 63) | (k << 6));
    }
    L.st[3][k][tid] = __int_as_float((want ? 2 : 0) | (slot << 2));
    cnt += __builtin_popcountll(m);
}
template <int K>
__device__ __forceinline__ bool eval_pop(const SlowLds<K> &L, int k, float (&c)[4])
{
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    const int f = __float_as_int(L.st[3][k][tid]);
    const bool want = (f & 2) != 0;
    if (want) {
        const int slot = f >> 2;
        c[0] = L.q[wave][0][slot]; c[1] = L.q[wave][1][slot]; c[2] = L.q[wave][2][slot]; c[3] = L.q[wave][3][slot];
    }
    return want;
}
__device__ __forceinline__ float lane_fetch(float v, int src)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)__float_as_uint(v)));
}
__device__ __forceinline__ V3 lane_fetch(V3 v, int src) { return mk(lane_fetch(v.x, src), lane_fetch(v.y, src), lane_fetch(v.z, src)); }
// what ggx_eval_pdf / ggx_fresnel / ggx_G read of a closure (the iors only enter through eta2)
__device__ __forceinline__ Ggx ggx_fetch(const Ggx &g, int src)
{
    Ggx h;
    h.fr.N = lane_fetch(g.fr.N, src); h.fr.U = lane_fetch(g.fr.U, src); h.fr.V = lane_fetch(g.fr.V, src);
    h.view = lane_fetch(g.view, src);
    h.ksR = lane_fetch(g.ksR, src); h.ksG = lane_fetch(g.ksG, src); h.ksB = lane_fetch(g.ksB, src);
    h.rough = lane_fetch(g.rough, src); h.ax = lane_fetch(g.ax, src); h.ay = lane_fetch(g.ay, src);
    h.iorIn = 0.0f; h.iorOut = 0.0f; h.etaIO = 0.0f;
    h.eta2 = lane_fetch(g.eta2, src); h.vn = lane_fetch(g.vn, src); h.g1v = lane_fetch(g.g1v, src);
    return h;
}
// the light-sampling strategy's evaluation (one light sample, both lobes) for the queued requests: per request the four
// terms f_r w / p, f_g w / p, f_b w / p (GGX) and f_d w_d / p (Oren-Nayar) of ggx_direct_loops.  Whole wavefront.
template <int K>
__device__ __forceinline__ void ggx_light_eval_run(SlowLds<K> &Q, int cnt, const Ggx &g, const OrenNayar &on, float conePdf,
                                                   bool sampleDiffuse, int mode)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    wave_lds_fence();
    for (int base = 0; base < cnt; base += 64) {
        const int j = base + lane;
        const bool have = j < cnt;
        V3 L = mk(0.0f, 0.0f, 1.0f);
        int src = lane;
        if (have) { L = mk(Q.q[wave][0][j], Q.q[wave][1][j], Q.q[wave][2][j]); src = __float_as_int(Q.q[wave][3][j]) & 63; }
        const Ggx h = ggx_fetch(g, src);                        // every lane executes the fetches
        OrenNayar o;
        o.N = h.fr.N; o.A = lane_fetch(on.A, src); o.B = lane_fetch(on.B, src);
        const float cp = lane_fetch(conePdf, src);
        const bool sd = lane_fetch(sampleDiffuse ? 1.0f : 0.0f, src) != 0.0f;
        if (have) {
            float fr, fg, fb, pb;
            ggx_eval_pdf<true, true>(h, L, fr, fg, fb, pb);
            const float wgt = mode == RLS_MIS_LIGHT_ONLY ? 1.0f : power_heuristic(cp, pb);
            float cA = 0.0f;
            if (sd) {
                const float fd = oren_nayar_brdf(o, h.view, L);
                const float wd = mode == RLS_MIS_LIGHT_ONLY ? 1.0f : power_heuristic(cp, oren_nayar_pdf(o, L));
                cA = R_DIV(fd * wd, cp);
            }
            Q.q[wave][0][j] = R_DIV(fr * wgt, cp); Q.q[wave][1][j] = R_DIV(fg * wgt, cp); Q.q[wave][2][j] = R_DIV(fb * wgt, cp);
            Q.q[wave][3][j] = cA;
        }
    }
    wave_lds_fence();
}

// the BSDF-sampling strategy's evaluation of the GGX samples that hit the light: f w / p_b per channel
template <int K>
__device__ __forceinline__ void ggx_hit_eval_run(SlowLds<K> &Q, int cnt, const Ggx &g, float conePdf, int mode)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    wave_lds_fence();
    for (int base = 0; base < cnt; base += 64) {
        const int j = base + lane;
        const bool have = j < cnt;
        V3 L = mk(0.0f, 0.0f, 1.0f);
        int src = lane;
        if (have) { L = mk(Q.q[wave][0][j], Q.q[wave][1][j], Q.q[wave][2][j]); src = __float_as_int(Q.q[wave][3][j]) & 63; }
        const Ggx h = ggx_fetch(g, src);
        const float cp = lane_fetch(conePdf, src);
        if (have) {
            float fr, fg, fb, pb;
            ggx_eval_pdf<true, true>(h, L, fr, fg, fb, pb);
            const float wgt = mode == RLS_MIS_BSDF_ONLY ? 1.0f : power_heuristic(pb, cp);
            Q.q[wave][0][j] = R_DIV(fr * wgt, pb); Q.q[wave][1][j] = R_DIV(fg * wgt, pb); Q.q[wave][2][j] = R_DIV(fb * wgt, pb);
        }
    }
    wave_lds_fence();
}

// integrateGlossy's sample loop over one closure (src/rlGgx.h:172-179 -> AiBRDFIntegrate over the triple): lane `sub`
// of a G-lane group takes samples sub, sub + G, ...; sums of f/pdf and of the Fresnel side effect of evalSample
// (src/rlGgx.h:103), in sample order and replicated over the group (fold)
// PACK = false: the plain loop.  rlSkin runs its two lobes inside per-point branches (src/rlSkin.cpp:191,214): wavefronts
// arrive here partly active, and the packed form costs more than it saves there (+17 % on the whole kernel, measured)
template <int G, int K, bool PACK = true>
__device__ __forceinline__ void ggx_glossy_loop(SlowLds<K> &slow, const Ggx &g, const VndfView &w,
                                                const uint32_t (*tab)[kMaxSpp], int spp, int sub, uint32_t sx, uint32_t sy,
                                                float &accR, float &accG, float &accB, float &accF, float f0 = 0.0f)
{
    // f0: the Fresnel sum of the samples drawn on the closure before (rlSkin's light loops)
    accR = 0.0f; accG = 0.0f; accB = 0.0f; accF = f0;
    if (!PACK) {
        for (int s0 = 0; s0 < spp; s0 += G) {                  // one round: G consecutive samples, one per lane
            const int s = s0 + sub;
            float tR = 0.0f, tG = 0.0f, tB = 0.0f, tF = 0.0f;
            if (s < spp) {
                float rx = bits_u01(tab[0][s] ^ sx);
                float ry = bits_u01(tab[1][s] ^ sy);
                V3 M = vndf_microfacet(w, g.fr, rx, ry);
                V3 L = reflect_direction(g.view, M);
                tF = ggx_fresnel(g, L, M);                      // mReflectWeight, src/rlGgx.h:103
                float fr, fg, fb, pdf;
                ggx_eval_pdf<true, true>(g, L, fr, fg, fb, pdf);
                tR = fr / pdf; tG = fg / pdf; tB = fb / pdf;
            }
            fold<G>(accF, tF); fold<G>(accR, tR); fold<G>(accG, tG); fold<G>(accB, tB);
        }
    }
    for (int s0 = sub; PACK && s0 - sub < spp; s0 += K * G) {   // K samples per pass (SlowLds)
        int cnt = 0;
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const int s = s0 + k * G;
            const int sc = s < spp ? s : 0;
            ggx_vndf_push<K>(slow, k, cnt, s < spp, w, bits_u01(tab[0][sc] ^ sx), bits_u01(tab[1][sc] ^ sy));
        }
        slow_run<K>(slow, cnt);
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            float tR = 0.0f, tG = 0.0f, tB = 0.0f, tF = 0.0f;
            if (s0 + k * G < spp) {
                V3 M = ggx_vndf_pop<K>(slow, k, w, g.fr);
                V3 L = reflect_direction(g.view, M);
                tF = ggx_fresnel(g, L, M);                      // mReflectWeight, src/rlGgx.h:103
                float fr, fg, fb, pdf;
                ggx_eval_pdf<true, true>(g, L, fr, fg, fb, pdf);
                tR = fr / pdf; tG = fg / pdf; tB = fb / pdf;
            }
            fold<G>(accF, tF); fold<G>(accR, tR); fold<G>(accG, tG); fold<G>(accB, tB);
        }
    }
}

// One light of a light loop, as the kernels below read it from the argument struct (l is wave-uniform)
struct LightRegs { int mode; float rad[3]; LightCone cone; };
__device__ __forceinline__ LightRegs light_regs(const rls_sphere_light &lt, V3 P)
{
    LightRegs r;
    r.mode = lt.mis_mode;
    r.rad[0] = lt.radiance[0]; r.rad[1] = lt.radiance[1]; r.rad[2] = lt.radiance[2];
    r.cone = cone_make(arr3(lt.center), lt.radius, P);
    return r;
}

// The light loop of one GGX lobe of rlSkin (src/rlSkin.cpp:193-198 / 217-222): per light evalLightSample
// (src/rlGgx.h:167-170) = the two-sample estimator of rls_ggx_direct_lighting's specular lobe.  out: the sum over the
// lights; f / cnt: the running Fresnel sum and the count of the evalSample calls (src/rlGgx.h:103) -- the caller carries
// f into integrateGlossy's loop, which goes on adding to it in sample order.  All sums are replicated over the lanes of
// the group (fold).  Sample streams: `stream` + 4 l (light samples), `stream` + 1 + 4 l (BSDF samples).
template <int G, class IO>
__device__ __forceinline__ void ggx_light_loops(const Ggx &g, const VndfView &w, V3 N, V3 P, const IO &io,
                                                const uint32_t (*tab)[kMaxSpp], int spp, int sub, float inv,
                                                uint32_t seed, uint64_t index, uint32_t stream,
                                                float out[3], float &f, float &cnt)
{
    out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f; f = 0.0f; cnt = 0.0f;
    for (int l = 0; l < io.nl; l++) {
        const LightRegs lt = light_regs(io.lights[l], P);   // io: the kernel's argument struct (scalar loads)
        const LightCone &cone = lt.cone;
        const int mode = lt.mode;
        uint32_t scr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) scr[k] = hash_u32(seed, index, kScrambleStream + 2 * (stream + 4 * l) + k);
        float sR = 0.0f, sG = 0.0f, sB = 0.0f;
        for (int s0 = 0; s0 < spp && cone.valid; s0 += G) {     // the plain loop: see ggx_glossy_loop, PACK = false
            const int s = s0 + sub;
            const bool ok = s < spp;
            float aR = 0.0f, aG = 0.0f, aB = 0.0f, bR = 0.0f, bG = 0.0f, bB = 0.0f, tF = 0.0f, tC = 0.0f;
            if (ok && mode != RLS_MIS_BSDF_ONLY) {
                float rx = bits_u01(tab[0][s] ^ scr[0]), ry = bits_u01(tab[1][s] ^ scr[1]);
                V3 L = cone_sample(cone, rx, ry);
                if (dot(L, N) > 0.0f) {
                    float fr, fg, fb, pb;
                    ggx_eval_pdf<true, true>(g, L, fr, fg, fb, pb);
                    float wgt = mode == RLS_MIS_LIGHT_ONLY ? 1.0f : power_heuristic(cone.pdf, pb);
                    aR = R_DIV(fr * wgt, cone.pdf); aG = R_DIV(fg * wgt, cone.pdf); aB = R_DIV(fb * wgt, cone.pdf);
                }
            }
            if (ok && mode != RLS_MIS_LIGHT_ONLY) {
                float rx = bits_u01(tab[0][s] ^ scr[2]), ry = bits_u01(tab[1][s] ^ scr[3]);
                V3 M = vndf_microfacet(w, g.fr, rx, ry);
                V3 L = reflect_direction(g.view, M);
                tF = ggx_fresnel(g, L, M);                      // mReflectWeight += ..., mMisSampleCount += 1
                tC = 1.0f;
                if (!is_zero(L) && dot(L, N) > 0.0f && cone_hit(cone, L)) {
                    float fr, fg, fb, pb;
                    ggx_eval_pdf<true, true>(g, L, fr, fg, fb, pb);
                    float wgt = mode == RLS_MIS_BSDF_ONLY ? 1.0f : power_heuristic(pb, cone.pdf);
                    bR = R_DIV(fr * wgt, pb); bG = R_DIV(fg * wgt, pb); bB = R_DIV(fb * wgt, pb);
                }
            }
            // one running sum per channel: the light sample's term, then the BSDF sample's, sample by sample
            fold2<G>(sR, aR, bR); fold2<G>(sG, aG, bG); fold2<G>(sB, aB, bB);
            fold<G>(f, tF);
            cnt += G == 1 ? tC : group_sum<G>(tC);
        }
        out[0] += lt.rad[0] * sR * inv; out[1] += lt.rad[1] * sG * inv; out[2] += lt.rad[2] * sB * inv;
    }
}

// ---------------------------------------------------------------------------------------------
using rlsh::GgxIntIO;
using rlsh::DisneyIntIO;


// ---------------------------------------------------------------------------------------------


// ---------------------------------------------------------------------------------------------
// SssSampler::integrateScatter over an analytic scene (src/rlSss.h:167-280, 293-356, 361-424,
// 439-454; include/rlshaders_amd.h, rls_sss_integrate_scatter, says what stands in for the closed
// renderer).  The reference shades every hit of a probe ray first and combines them afterwards;
// hit k's combination only adds to the running sums, so shading and combining hit by hit gives
// the same sums in the same order.
using rlsh::ScatterIO;


__device__ __forceinline__ NdProfile scatter_profile(const rls_sss_closure &c, const PIndex<int64_t> &k)
{
    float m = ldp(c.sss_dist_multiplier, k);   // src/rlSkin.cpp:235-236
    return nd_make<true>(ldp(c.sss_scatter_dist[0], k) * m, ldp(c.sss_scatter_dist[1], k) * m,
                   ldp(c.sss_scatter_dist[2], k) * m);
}

// the analytic scene in registers
struct SceneRegs {
    bool sphere, has_gate, cavity, literal;
    V3 planeN, planeP, center, Ldir, gateP, gateN;
    float radius, lc[3];
};
__device__ __forceinline__ SceneRegs scene_regs(const rls_sss_scene &sc)
{
    SceneRegs r;
    r.sphere = sc.geometry == RLS_SCENE_SPHERE;
    r.has_gate = sc.has_gate != 0; r.cavity = sc.use_cavity_fade != 0; r.literal = sc.literal_matrix != 0;
    r.planeN = arr3(sc.plane_normal); r.planeP = arr3(sc.plane_point); r.center = arr3(sc.sphere_center);
    r.Ldir = arr3(sc.light_dir); r.gateP = arr3(sc.gate_point); r.gateN = arr3(sc.gate_normal);
    r.radius = sc.sphere_radius;
    r.lc[0] = sc.light_color[0]; r.lc[1] = sc.light_color[1]; r.lc[2] = sc.light_color[2];
    return r;
}

// the probe-ray loop of integrateScatter (src/rlSss.h:224-270) for one shading point: sums of irradiance / pdf (in sample
// order, replicated over the G-lane group: fold) and of the shaded-hit count over the samples sub, sub + G, ...
template <int G>
__device__ __forceinline__ void scatter_loop(const NdProfile &p, const Frame &fr, V3 Po, const SceneRegs &sc,
                                             const uint32_t (*tab)[kMaxSpp], int spp, int sub, uint32_t sx, uint32_t sy,
                                             float &accR, float &accG, float &accB, float &accD)
{
    accR = 0.0f; accG = 0.0f; accB = 0.0f; accD = 0.0f;
    for (int s0 = 0; s0 < spp; s0 += G) {                      // one round: G consecutive samples, one per lane
        const int s = s0 + sub;
        float tr[2][3] = { { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f } };       // the terms of this sample's (up to) two hits
        if (s < spp) {
        float rx = bits_u01(tab[0][s] ^ sx);
        float ry = bits_u01(tab[1][s] ^ sy);
        V3 off, dir;
        float maxdist;
        sss_probe_ray(p, fr, rx, ry, off, dir, maxdist);                     // :228
        const V3 O = Po + off;
        // AiTraceProbe (:293): the roots of the ray against the plane / sphere, ascending
        float cand[2];
        bool has[2] = { false, false };
        if (sc.sphere) {
            V3 oc = O - sc.center;
            float qa = dot(dir, dir);
            float qb = dot(oc, dir);
            float qc = dot(oc, oc) - sc.radius * sc.radius;
            float disc = qb * qb - qa * qc;
            if (!(disc < 0.0f) && qa != 0.0f) {
                float sq = R_SQRT(disc);
                cand[0] = R_DIV(-qb - sq, qa);
                cand[1] = R_DIV(-qb + sq, qa);
                has[0] = has[1] = true;
            }
        } else {
            float denom = dot(sc.planeN, dir);
            if (denom != 0.0f) {
                cand[0] = R_DIV(dot(sc.planeN, sc.planeP - O), denom);
                has[0] = true;
            }
        }
        V3 prev = Po;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!has[k] || !(cand[k] > 0.0f && cand[k] <= maxdist)) continue;
            const V3 hp = O + dir * cand[k];
            const V3 hn = sc.sphere ? normalize(hp - sc.center) : sc.planeN;
            if (!(length(prev - hp) > kEps)) continue;                       // :316-317
            prev = hp;
            // shadeProbeSample, :379-420
            const V3 d = hp - Po;
            const float r = length(d);
            if (r > p.maxR) continue;
            float fade = 1.0f;
            if (sc.cavity) fade = sss_cavity_fade(d, r, hn, fr.N);
            if (!(fade > kEps)) continue;
            accD += 1.0f;
            // evalLightSample, :439-454
            float w = kInvPi * maxf(0.0f, dot(hn, sc.Ldir));
            if (sc.has_gate && !(dot(hp - sc.gateP, sc.gateN) > 0.0f)) w = 0.0f;
            float pr, pg, pb;
            nd_profile(p, r, pr, pg, pb);
            const float iR = sc.lc[0] * w * pr * fade;
            const float iG = sc.lc[1] * w * pg * fade;
            const float iB = sc.lc[2] * w * pb * fade;
            if (iR == 0.0f && iG == 0.0f && iB == 0.0f) continue;            // :249
            const float pdf = sss_mis_pdf(p, fr, d, hn, sc.literal);
            tr[k][0] = R_DIV(iR, pdf); tr[k][1] = R_DIV(iG, pdf); tr[k][2] = R_DIV(iB, pdf);
        }
        }
        // hit by hit, sample by sample: the order the one-lane loop adds in
        fold2<G>(accR, tr[0][0], tr[1][0]); fold2<G>(accG, tr[0][1], tr[1][1]); fold2<G>(accB, tr[0][2], tr[1][2]);
    }
    if (G > 1) accD = group_sum<G>(accD);                       // shaded-hit count: integers, any order
}


// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// integrateRefract (src/rlGgx.h:205-245).  Traced branch (228-244): per sample a microfacet normal, the refraction
// of the view about it (the mirror direction on total internal reflection), radiance x getSampleWeight, the sum
// x AiSamplerGetSampleInvCount.  Untraced branch (213-222): one refraction about the shading normal, radiance x
// SQR(iorOut / iorIn) x |Nf . dir|, black on total internal reflection.  AiTrace / AiTraceBackground are closed: the
// radiance is that of a uniform environment, `env` (parity unpinned).
using rlsh::RefractIntIO;

// the traced branch's sample loop (src/rlGgx.h:228-244): mean sample weight and fraction of total internal reflections
template <int G, int K>
__device__ __forceinline__ void ggx_refract_loop(SlowLds<K> &slow, const Ggx &g, const VndfView &w,
                                                 const uint32_t (*tab)[kMaxSpp], int spp, int sub, uint32_t sx, uint32_t sy,
                                                 float &acc, float &tir)
{
    acc = 0.0f; tir = 0.0f;
    for (int s0 = sub; s0 - sub < spp; s0 += K * G) {           // K samples per pass (SlowLds)
        int cnt = 0;
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const int s = s0 + k * G;
            const int sc = s < spp ? s : 0;
            ggx_vndf_push<K>(slow, k, cnt, s < spp, w, bits_u01(tab[0][sc] ^ sx), bits_u01(tab[1][sc] ^ sy));
        }
        slow_run<K>(slow, cnt);
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            float t = 0.0f;
            if (s0 + k * G < spp) {
                V3 M = ggx_vndf_pop<K>(slow, k, w, g.fr);
                V3 dir;
                if (!ggx_refract(g, M, dir)) tir += 1.0f;
                t = ggx_sample_weight(g, g.view, dir, M);                // :241
            }
            fold<G>(acc, t);
        }
    }
    if (G > 1) tir = group_sum<G>(tir);                             // a count: integers, any order
    const float inv = 1.0f / (float)spp;                             // AiSamplerGetSampleInvCount, :244
    acc *= inv; tir *= inv;
}
// the untraced branch (213-222): one refraction about the shading normal
__device__ __forceinline__ void ggx_refract_untraced(const Ggx &g, float &acc, float &tir)
{
    acc = 0.0f; tir = 0.0f;
    V3 dir;
    if (ggx_refract(g, g.fr.N, dir)) acc = g.eta2 * absf(dot(g.fr.N, dir));   // :216
    else tir = 1.0f;
}


// ---------------------------------------------------------------------------------------------
// Direct lighting of the rlGgx node (src/rlGgx.cpp:274-299); include/rlshaders_amd.h,
// rls_ggx_direct_lighting, says what stands in for the closed light loop.
using rlsh::LightIO;

// The light loop of rlGgx (src/rlGgx.cpp:285-299) for one shading point: per light one AiEvaluateLightSample over the
// Oren-Nayar closure (when sampleDiffuse) and one over the GGX triple; oD / oS = the sums over the lights, group-reduced,
// BEFORE `diffuse *= diffuseColor; specular *= specularWeight` (304-305).  Light l: sample streams 3 l .. 3 l + 2.
template <int G, int K, class IO>
__device__ __forceinline__ void ggx_direct_loops(SlowLds<K> &slow, const Ggx &g, const VndfView &w, const OrenNayar &on, V3 wo, V3 N, V3 P,
                                                 bool sampleDiffuse, const IO &io,
                                                 const uint32_t (*tab)[kMaxSpp], int spp, int sub, float inv,
                                                 uint32_t seed, uint64_t index, float oD[3], float oS[3])
{
    oS[0] = 0.0f; oS[1] = 0.0f; oS[2] = 0.0f; oD[0] = 0.0f; oD[1] = 0.0f; oD[2] = 0.0f;
    for (int l = 0; l < io.nl; l++) {                          // while (AiLightsGetSample(sg)), src/rlGgx.cpp:286
        const LightRegs lt = light_regs(io.lights[l], P);      // io: the kernel's argument struct (scalar loads)
        const LightCone &cone = lt.cone;
        const int mode = lt.mode;
        uint32_t scr[6];
#pragma unroll
        for (int k = 0; k < 6; k++) scr[k] = hash_u32(seed, index, kScrambleStream + 6 * l + k);

        // The estimator's two strategies as separate passes over the samples, each with its own sums (grown in sample order,
        // added at the end): light samples first, then BSDF samples.
        float lR = 0.0f, lG = 0.0f, lB = 0.0f, lA = 0.0f, bR = 0.0f, bG = 0.0f, bB = 0.0f, bA = 0.0f;
        for (int s0 = sub; mode != RLS_MIS_BSDF_ONLY && s0 - sub < spp; s0 += K * G) {   // one light sample, both lobes
            // the samples above the horizon are queued and evaluated packed (eval_push / ggx_light_eval_run / eval_pop)
            int qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const int sc = s < spp ? s : 0;
                V3 L = cone_sample(cone, bits_u01(tab[0][sc] ^ scr[0]), bits_u01(tab[1][sc] ^ scr[1]));
                eval_push<K>(slow, k, qn, s < spp && cone.valid && dot(L, N) > 0.0f, L);
            }
            ggx_light_eval_run<K>(slow, qn, g, on, cone.pdf, sampleDiffuse, mode);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                float t[4], u[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                if (eval_pop<K>(slow, k, t)) {
                    u[0] = t[0]; u[1] = t[1]; u[2] = t[2];
                    if (sampleDiffuse) u[3] = t[3];
                }
                fold<G>(lR, u[0]); fold<G>(lG, u[1]); fold<G>(lB, u[2]); fold<G>(lA, u[3]);
            }
        }
        for (int s0 = sub; mode != RLS_MIS_LIGHT_ONLY && s0 - sub < spp; s0 += K * G) {   // one BSDF sample per lobe; K per pass (SlowLds)
            int qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const int sc = s < spp ? s : 0;
                ggx_vndf_push<K>(slow, k, qn, s < spp && cone.valid, w, bits_u01(tab[0][sc] ^ scr[2]),
                                 bits_u01(tab[1][sc] ^ scr[3]));
            }
            slow_run<K>(slow, qn);
            // the reflected directions; the few that hit the light are queued for evaluation (the queue is free again
            // once every sample's slopes have been picked up)
            uint32_t hits = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                V3 M = ggx_vndf_pop<K>(slow, k, w, g.fr);
                V3 L = reflect_direction(g.view, M);
                const bool hit = s < spp && cone.valid && !is_zero(L) && dot(L, N) > 0.0f && cone_hit(cone, L);
                hits |= (hit ? 1u : 0u) << k;
                slow.st[0][k][threadIdx.x] = L.x; slow.st[1][k][threadIdx.x] = L.y; slow.st[2][k][threadIdx.x] = L.z;
            }
            wave_lds_fence();
            qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const V3 L = mk(slow.st[0][k][threadIdx.x], slow.st[1][k][threadIdx.x], slow.st[2][k][threadIdx.x]);
                eval_push<K>(slow, k, qn, ((hits >> k) & 1u) != 0, L);
            }
            ggx_hit_eval_run<K>(slow, qn, g, cone.pdf, mode);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                float u[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                if (s < spp && cone.valid) {
                    float t[4];
                    if (eval_pop<K>(slow, k, t)) { u[0] = t[0]; u[1] = t[1]; u[2] = t[2]; }
                    if (sampleDiffuse) {
                        float rx = bits_u01(tab[0][s] ^ scr[4]), ry = bits_u01(tab[1][s] ^ scr[5]);
                        V3 Ld = cosine_hemisphere(g.fr, rx, ry);
                        float pd = oren_nayar_pdf(on, Ld);
                        if (pd > 0.0f && cone_hit(cone, Ld)) {
                            float fd = oren_nayar_brdf(on, wo, Ld);
                            float wd = mode == RLS_MIS_BSDF_ONLY ? 1.0f : power_heuristic(pd, cone.pdf);
                            u[3] = R_DIV(fd * wd, pd);
                        }
                    }
                }
                fold<G>(bR, u[0]); fold<G>(bG, u[1]); fold<G>(bB, u[2]); fold<G>(bA, u[3]);
            }
        }
        const float sR = lR + bR, sG = lG + bG, sB = lB + bB, dA = lA + bA;
        // specular += ..., diffuse += ... (288-294); the first light assigns (0 + x loses the sign of a zero)
        const float tS[3] = { lt.rad[0] * sR * inv, lt.rad[1] * sG * inv, lt.rad[2] * sB * inv };
        const float tD[3] = { lt.rad[0] * dA * inv, lt.rad[1] * dA * inv, lt.rad[2] * dA * inv };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            oS[k] = l == 0 ? tS[k] : oS[k] + tS[k];
            oD[k] = l == 0 ? tD[k] : oD[k] + tD[k];
        }
    }
}

__device__ __forceinline__ bool color_is_small(float r, float g, float b)      // AiColorIsSmall
{
    return absf(r) < kEps && absf(g) < kEps && absf(b) < kEps;
}


// ---------------------------------------------------------------------------------------------
// Direct lighting of the rlDisney node (src/rlDisney.cpp:695-705): per light the diffuse lobe's and the specular
// lobe's AiEvaluateLightSample over the callback triple (265-277); include/rlshaders_amd.h,
// rls_disney_direct_lighting, says what stands in for the closed light loop.
using rlsh::DisneyLightIO;

// the closure of shading point ii, prepared (a macro: the same lines in a function cost 26 more spilled registers)
#define RLS_DISNEY_LOAD(d, c, ii)                                                                          \
    Disney d;                                                                                              \
    {                                                                                                      \
        float br_, bg_, bb_, sc_[10];                                                                      \
        const PIndex<int64_t> pk = pindex((c).materials, (int64_t)(ii));                                   \
        ldrgb((c).base_color, pk, br_, bg_, bb_);                                                          \
        sc_[0] = ldp((c).subsurface, pk); sc_[1] = ldp((c).metallic, pk); sc_[2] = ldp((c).specular, pk);  \
        sc_[3] = ldp((c).specular_tint, pk); sc_[4] = ldp((c).roughness, pk); sc_[5] = ldp((c).anisotropic, pk); \
        sc_[6] = ldp((c).sheen, pk); sc_[7] = ldp((c).sheen_tint, pk); sc_[8] = ldp((c).clearcoat, pk);    \
        sc_[9] = ldp((c).clearcoat_gloss, pk);                                                             \
        d = disney_make(ld3((c).wo, ii), ld3((c).N, ii), ld3((c).T, ii), br_, bg_, bb_, sc_);              \
        disney_prepare(d);                                                                                 \
    }

// what disney_eval_pdf reads of a prepared closure
__device__ __forceinline__ Disney disney_fetch(const Disney &d, int src)
{
    Disney h;
    h.fr.N = lane_fetch(d.fr.N, src); h.fr.U = lane_fetch(d.fr.U, src); h.fr.V = lane_fetch(d.fr.V, src);
    h.view = lane_fetch(d.view, src);
    h.f0R = lane_fetch(d.f0R, src); h.f0G = lane_fetch(d.f0G, src); h.f0B = lane_fetch(d.f0B, src);
    h.shR = lane_fetch(d.shR, src); h.shG = lane_fetch(d.shG, src); h.shB = lane_fetch(d.shB, src);
    h.baseR = lane_fetch(d.baseR, src); h.baseG = lane_fetch(d.baseG, src); h.baseB = lane_fetch(d.baseB, src);
    h.roughness = lane_fetch(d.roughness, src); h.subsurface = lane_fetch(d.subsurface, src);
    h.metallic = 0.0f; h.clearcoatGloss = 0.0f; h.gtr2Weight = 0.0f;          // not read by the evaluation
    h.clearcoat = lane_fetch(d.clearcoat, src); h.specRough = lane_fetch(d.specRough, src);
    h.ax = lane_fetch(d.ax, src); h.ay = lane_fetch(d.ay, src);
    h.vn = lane_fetch(d.vn, src); h.FV = lane_fetch(d.FV, src); h.gsV = lane_fetch(d.gsV, src); h.grV = lane_fetch(d.grV, src);
    h.ccA2m1 = lane_fetch(d.ccA2m1, src); h.ccLogA2 = lane_fetch(d.ccLogA2, src);
    h.ccw = lane_fetch(d.ccw, src); h.vnc = lane_fetch(d.vnc, src); h.om = lane_fetch(d.om, src);
#if !RLS_FAST
    h.yax = 0.0f; h.yay = 0.0f;      // the reciprocals of alpha_x, alpha_y stay at home: a fetched closure divides the IEEE way
    h.yW = 0.0f; h.y1mW = 0.0f;
#endif
    return h;
}
// the light-sampling strategy of rlDisney's light loop for the queued light samples: both lobes (evalDiffuseLightSample,
// evalSpecularLightSample); the diffuse lobe's three terms go back through the queue, the specular lobe's through the
// requesting lane's state words st[0..2][k]
template <int K>
__device__ __forceinline__ void disney_light_eval_run(SlowLds<K> &Q, int cnt, const Disney &d, float conePdf, int mode)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    wave_lds_fence();
    for (int base = 0; base < cnt; base += 64) {
        const int j = base + lane;
        const bool have = j < cnt;
        V3 L = mk(0.0f, 0.0f, 1.0f);
        int who = lane;
        if (have) { L = mk(Q.q[wave][0][j], Q.q[wave][1][j], Q.q[wave][2][j]); who = __float_as_int(Q.q[wave][3][j]); }
        const int src = who & 63, k = who >> 6;
        const Disney h = disney_fetch(d, src);
        const float cp = lane_fetch(conePdf, src);
        if (have) {
            float r, g, b, p;
            disney_eval_pdf<true, true, true>(h, L, r, g, b, p);       // evalDiffuseLightSample, src/rlDisney.cpp:265-269
            float wgt = mode == RLS_MIS_LIGHT_ONLY ? 1.0f : power_heuristic(cp, p);
            Q.q[wave][0][j] = R_DIV(r * wgt, cp); Q.q[wave][1][j] = R_DIV(g * wgt, cp); Q.q[wave][2][j] = R_DIV(b * wgt, cp);
            disney_eval_pdf<false, true, true>(h, L, r, g, b, p);      // evalSpecularLightSample, :272-276
            wgt = mode == RLS_MIS_LIGHT_ONLY ? 1.0f : power_heuristic(cp, p);
            const int t = (wave << 6) | src;
            Q.st[0][k][t] = R_DIV(r * wgt, cp); Q.st[1][k][t] = R_DIV(g * wgt, cp); Q.st[2][k][t] = R_DIV(b * wgt, cp);
        }
    }
    wave_lds_fence();
}
// the BSDF-sampling strategy for the queued samples that hit the light: f w / p per channel and, in the fourth word,
// whether the sample counts (pdf > AI_EPSILON, src/rlDisney.cpp:309)
template <int K, bool DIFFUSE>
__device__ __forceinline__ void disney_hit_eval_run(SlowLds<K> &Q, int cnt, const Disney &d, float conePdf, int mode)
{
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    wave_lds_fence();
    for (int base = 0; base < cnt; base += 64) {
        const int j = base + lane;
        const bool have = j < cnt;
        V3 L = mk(0.0f, 0.0f, 1.0f);
        int src = lane;
        if (have) { L = mk(Q.q[wave][0][j], Q.q[wave][1][j], Q.q[wave][2][j]); src = __float_as_int(Q.q[wave][3][j]) & 63; }
        const Disney h = disney_fetch(d, src);
        const float cp = lane_fetch(conePdf, src);
        if (have) {
            float r, g, b, p;
            disney_eval_pdf<DIFFUSE, true, true>(h, L, r, g, b, p);
            const float wgt = mode == RLS_MIS_BSDF_ONLY ? 1.0f : power_heuristic(p, cp);
            Q.q[wave][0][j] = R_DIV(r * wgt, p); Q.q[wave][1][j] = R_DIV(g * wgt, p); Q.q[wave][2][j] = R_DIV(b * wgt, p);
            Q.q[wave][3][j] = p > kEps ? 1.0f : 0.0f;
        }
    }
    wave_lds_fence();
}

// The light loop of rlDisney (src/rlDisney.cpp:695-705) for one shading point: oD / oS = the sums over the lights of
// evalDiffuseLightSample / evalSpecularLightSample, group-reduced.  Light l: sample streams 3 l .. 3 l + 2.
template <int G, int K, class IO>
__device__ __forceinline__ void disney_direct_loops(SlowLds<K> &slow, const Disney &d, const VndfView &w, V3 N, V3 P, const IO &io,
                                                    const uint32_t (*tab)[kMaxSpp], int spp, int sub, float inv,
                                                    uint32_t seed, uint64_t index, float oD[3], float oS[3])
{
    oS[0] = 0.0f; oS[1] = 0.0f; oS[2] = 0.0f; oD[0] = 0.0f; oD[1] = 0.0f; oD[2] = 0.0f;
    for (int l = 0; l < io.nl; l++) {                          // while (AiLightsGetSample(sg)), :696
        const LightRegs lt = light_regs(io.lights[l], P);
        const LightCone &cone = lt.cone;
        const int mode = lt.mode;
        uint32_t scr[6];
#pragma unroll
        for (int k = 0; k < 6; k++) scr[k] = hash_u32(seed, index, kScrambleStream + 6 * l + k);

        // The estimator's two strategies as separate passes over the samples (K per pass, the same trip count in every
        // lane), each with its own sums, grown in sample order and added at the end.  Every evaluation is queued and run
        // packed: a light sample is evaluated only above the horizon, a BSDF sample only where it hits the light.
        float lD[3] = { 0.0f, 0.0f, 0.0f }, lS[3] = { 0.0f, 0.0f, 0.0f }, bD[3] = { 0.0f, 0.0f, 0.0f }, bS[3] = { 0.0f, 0.0f, 0.0f };
        const int tid = (int)threadIdx.x;
        for (int s0 = sub; mode != RLS_MIS_BSDF_ONLY && s0 - sub < spp; s0 += K * G) {     // one light sample, both lobes
            int qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const int sc = s < spp ? s : 0;
                V3 L = cone_sample(cone, bits_u01(tab[0][sc] ^ scr[0]), bits_u01(tab[1][sc] ^ scr[1]));
                eval_push<K>(slow, k, qn, s < spp && cone.valid && dot(L, N) > 0.0f, L);
            }
            disney_light_eval_run<K>(slow, qn, d, cone.pdf, mode);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                float t[4];
                float u[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
                if (eval_pop<K>(slow, k, t)) {
                    u[0] = t[0]; u[1] = t[1]; u[2] = t[2];
                    u[3] = slow.st[0][k][tid]; u[4] = slow.st[1][k][tid]; u[5] = slow.st[2][k][tid];
                }
                fold<G>(lD[0], u[0]); fold<G>(lD[1], u[1]); fold<G>(lD[2], u[2]);
                fold<G>(lS[0], u[3]); fold<G>(lS[1], u[4]); fold<G>(lS[2], u[5]);
            }
        }
        for (int s0 = sub; mode != RLS_MIS_LIGHT_ONLY && s0 - sub < spp; s0 += K * G) {    // one BSDF sample per lobe
            // diffuse lobe: cosine-weighted directions; those that hit the light are evaluated
            int qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const int sc = s < spp ? s : 0;
                V3 L = cosine_hemisphere(d.fr, bits_u01(tab[0][sc] ^ scr[2]), bits_u01(tab[1][sc] ^ scr[3]));
                eval_push<K>(slow, k, qn, s < spp && cone.valid && cone_hit(cone, L), L);
            }
            disney_hit_eval_run<K, true>(slow, qn, d, cone.pdf, mode);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                float t[4];
                float u[3] = { 0.0f, 0.0f, 0.0f };
                if (eval_pop<K>(slow, k, t) && t[3] != 0.0f) { u[0] = t[0]; u[1] = t[1]; u[2] = t[2]; }
                fold<G>(bD[0], u[0]); fold<G>(bD[1], u[1]); fold<G>(bD[2], u[2]);
            }
            // specular lobe: the sampler's rare branches packed, then the reflected directions that hit the light
            qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const int sc = s < spp ? s : 0;
                disney_spec_push<K>(slow, k, qn, s < spp && cone.valid, d, w, bits_u01(tab[0][sc] ^ scr[4]),
                                    bits_u01(tab[1][sc] ^ scr[5]));
            }
            slow_run<K>(slow, qn);
            uint32_t hits = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const V3 L = disney_spec_pop<K>(slow, k, d, w);
                const bool hit = s < spp && cone.valid && cone_hit(cone, L);
                hits |= (hit ? 1u : 0u) << k;
                slow.st[0][k][tid] = L.x; slow.st[1][k][tid] = L.y; slow.st[2][k][tid] = L.z;
            }
            wave_lds_fence();
            qn = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const V3 L = mk(slow.st[0][k][tid], slow.st[1][k][tid], slow.st[2][k][tid]);
                eval_push<K>(slow, k, qn, ((hits >> k) & 1u) != 0, L);
            }
            disney_hit_eval_run<K, false>(slow, qn, d, cone.pdf, mode);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                float t[4];
                float u[3] = { 0.0f, 0.0f, 0.0f };
                if (eval_pop<K>(slow, k, t) && t[3] != 0.0f) { u[0] = t[0]; u[1] = t[1]; u[2] = t[2]; }
                fold<G>(bS[0], u[0]); fold<G>(bS[1], u[1]); fold<G>(bS[2], u[2]);
            }
        }
        const float dR = lD[0] + bD[0], dG = lD[1] + bD[1], dB = lD[2] + bD[2];
        const float sR = lS[0] + bS[0], sG = lS[1] + bS[1], sB = lS[2] + bS[2];
        const float tD[3] = { lt.rad[0] * dR * inv, lt.rad[1] * dG * inv, lt.rad[2] * dB * inv };
        const float tS[3] = { lt.rad[0] * sR * inv, lt.rad[1] * sG * inv, lt.rad[2] * sB * inv };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            oD[k] = l == 0 ? tD[k] : oD[k] + tD[k];
            oS[k] = l == 0 ? tS[k] : oS[k] + tS[k];
        }
    }
}


// ---------------------------------------------------------------------------------------------

// lanes per point: fill >= ~4 waves per SIMD on every CU when the batch is small
int pick_group(const rls_context *ctx, int64_t n, int spp)
{
    if (const char *s = getenv("RLS_INTEGRATE_GROUP")) {
        int g = atoi(s);
        if (g == 1 || g == 4 || g == 16 || g == 64) return g;
    }
    const int64_t want_lanes = (int64_t)ctx->compute_units * 4 * 4 * 64;
    int g = 1;
    while (g < 64 && n * g < want_lanes && g * 4 <= spp) g *= 4;
    return g;
}

template <typename K, typename IO>
rls_status launch_g(rls_context *ctx, K k1, K k4, K k16, K k64, int g, const IO &io, const char *name)
{
    K k = g == 1 ? k1 : g == 4 ? k4 : g == 16 ? k16 : k64;
    dim3 grid = rlsh::grid_for(ctx, io.n, rlsh::kBlock / g);
    hipLaunchKernelGGL(k, grid, dim3(rlsh::kBlock), 0, ctx->stream, io);
    return rlsh::check_launch(name);
}

// the lights of a light loop, validated and copied into a kernel's argument struct
inline rls_status copy_lights(const rls_sphere_light *lights, int n_lights, int at_least, rls_sphere_light *dst, int *count)
{
    RLS_REQUIRE(n_lights >= at_least && n_lights <= RLS_MAX_LIGHTS, "n_lights out of range (RLS_MAX_LIGHTS)");
    RLS_REQUIRE(n_lights == 0 || lights != nullptr, "lights is NULL");
    for (int l = 0; l < n_lights; l++) {
        RLS_REQUIRE(lights[l].mis_mode >= RLS_MIS_BOTH && lights[l].mis_mode <= RLS_MIS_BSDF_ONLY, "unknown mis_mode");
        RLS_REQUIRE(lights[l].radius > 0.0f, "light radius must be positive");
        dst[l] = lights[l];
    }
    *count = n_lights;
    return RLS_OK;
}

// plane pointers advanced by k points (chunked / sharded calls)
inline rls_param adv(rls_param p, int64_t k) { if (p.v) p.v += k; return p; }
inline rls_param_rgb adv(rls_param_rgb p, int64_t k) { if (p.r) { p.r += k; p.g += k; p.b += k; } return p; }
inline rls_cvec3 adv(rls_cvec3 v, int64_t k) { v.x += k; v.y += k; v.z += k; return v; }
inline rls_rgb adv(rls_rgb v, int64_t k) { v.r += k; v.g += k; v.b += k; return v; }

} // namespace
