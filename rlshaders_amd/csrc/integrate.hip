// integrate.hip -- n^2-samples-per-point integrators for rlGgx and rlDisney: the per-sample loop
// arithmetic of the reference's glossy / diffuse integration (the loops Arnold's AiBRDFIntegrate
// runs over the callback triple, src/rlGgx.h:172-179, src/rlDisney.cpp:240-315; explicit form at
// src/rlDisney.cpp:299-312) with the random numbers drawn in-kernel.
//
// Sampler: stand-in for the closed AiSampler(n, 2) (src/rlGgx.cpp:148, src/rlDisney.cpp:68,72):
// a per-point XOR-scrambled (0,2)-sequence -- sample s has x = bitreverse(s) ^ scr_x,
// y = sobol2(s) ^ scr_y -- which is stratified on every elementary interval of the n^2 samples.
// The unscrambled table (2 x spp words) is staged in LDS once per workgroup; scrambles come from
// the counter hash of (seed, point index).
//
// Mapping: G lanes cooperate on one shading point (G = 1, 4, 16 or 64, chosen on the host from
// the batch size so that small batches still fill 256 CUs).  Lane `sub` of a group takes samples
// sub, sub+G, ...; the closure setup (frame, alphas, stretched-view analysis) is done once per
// lane and the partial sums are combined with wave64 butterfly shuffles.  G = 1 adds samples in
// ascending order, exactly like the reference's `result +=` loop.
//
// Roofline: fp32 VALU (not HBM) in reduced mode -- 88 B of parameters in and 32 B out per point
// against 2*n^2 triples of arithmetic (SURVEY.md section 8(d), config 3 mode R).  Streamed mode
// writes 28 B per triple; it too runs at the speed of its arithmetic.
//
// The same file holds the loops built on those: integrateRefract, the light loops of rlGgx / rlDisney
// (two-sample MIS over up to eight spherical lights), SssSampler::integrateScatter over an analytic
// scene, and the three nodes' whole shader_evaluate (rls_ggx_shade, rls_disney_shade,
// rls_skin_integrate).  The loops take K = 4 samples per pass and evaluate their samplers' rare
// branches packed through LDS (SlowLds below).
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
#ifndef RLS_INT_RELOAD
#define RLS_INT_RELOAD 1
#endif
#if RLS_INT_RELOAD
#define RLS_INT_ARGS(a) reload_args(a)
#else
#define RLS_INT_ARGS(a) (a)
#endif
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
    // rescaled rx and A = 2 rx' / G1 - 1 through their reciprocals (rlm::div32_y) unless some lane has none (RLS_LOOP_RECIP)
    const float y = gtr2 ? d.yW : d.y1mW;
    if (__builtin_expect(RLS_LOOP_RECIP && __builtin_amdgcn_ballot_w64(y == 0.0f || w.yG1 == 0.0f) == 0ull, 1)) {
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
    if (__builtin_expect(RLS_LOOP_RECIP && __builtin_amdgcn_ballot_w64(w.yG1 == 0.0f) == 0ull, 1))
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

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void ggx_integrate_kernel(GgxIntIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<RLS_SPEC_BLOCK> slow;
    stage_libm_tables();   // atanf range table (+ expf / logf / powf tables) -> LDS
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    // all lanes of a wave iterate the same number of times (shuffles need every lane live)
    const int64_t rounds = (a.n + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const rls_ggx_closure &c = a.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        V3 wo = ld3(c.wo, ii), N = ld3(c.N, ii), T = ld3(c.T, ii);
        float kr, kg, kb;
        ldrgb(c.KsColor, pk, kr, kg, kb);
        bool exiting = c.exiting ? (c.exiting[ii] != 0) : false;
        Ggx g = ggx_make(wo, N, T, exiting, kr, kg, kb, ldp(c.ior, pk), ldp(c.specularRoughness, pk),
                         ldp(c.anisotropic, pk));
        VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
        const uint32_t sx = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream);
        const uint32_t sy = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + 1);

        float accR, accG, accB, accF;
        ggx_glossy_loop<G>(slow, g, w, tab, a.spp, sub, sx, sy, accR, accG, accB, accF);
        if (live && sub == 0) {
            strgb(a.sum, i, accR, accG, accB);
            // getAvgReflectWeight, src/rlGgx.h:181-184
            stg(a.avgF, i, a.spp > 0 ? accF / (float)a.spp : 1.0f);
        }
    }
}

// ---------------------------------------------------------------------------------------------

// the kernel body: inlined into disney_integrate_kernel (the product) and disney_integrate_kernel_stamped (diagnostic: the
// same body between clock stamps, rls_internal.hpp ClockStamp).  `a` is the kernel's first parameter (reload_args).
template <int G, int FAST_MATH>
__device__ __forceinline__ void disney_integrate_body(const DisneyIntIO a)
{
    constexpr int K = RLS_SPEC_BLOCK;
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<K> slow;
    stage_libm_tables();   // powf / logf tables -> LDS (EXACT mode)
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        // the closure's 22 plane pointers re-read from the kernarg segment here, the 8 output pointers at the stores
        // (rls_internal.hpp, reload_args): none of them stays in a scalar register across the sample loop
        const DisneyIntIO al = RLS_INT_ARGS(a);
        const rls_disney_closure &c = al.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        V3 wo = ld3(c.wo, ii), N = ld3(c.N, ii), T = ld3(c.T, ii);
        float br, bg, bb;
        ldrgb(c.base_color, pk, br, bg, bb);
        float sc[10];
        sc[0] = ldp(c.subsurface, pk); sc[1] = ldp(c.metallic, pk); sc[2] = ldp(c.specular, pk);
        sc[3] = ldp(c.specular_tint, pk); sc[4] = ldp(c.roughness, pk); sc[5] = ldp(c.anisotropic, pk);
        sc[6] = ldp(c.sheen, pk); sc[7] = ldp(c.sheen_tint, pk); sc[8] = ldp(c.clearcoat, pk);
        sc[9] = ldp(c.clearcoat_gloss, pk);
        Disney d = disney_make(wo, N, T, br, bg, bb, sc);
        disney_prepare(d);        // what evalBrdf / evalPdf / evalSample recompute per call from the closure alone
        VndfView w = vndf_view(d.view, d.fr, d.ax, d.ay);
        const uint32_t dx = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream);
        const uint32_t dy = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + 1);
        const uint32_t sx = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + 2);
        const uint32_t sy = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + 3);

        float dR = 0.0f, dG = 0.0f, dB = 0.0f, dC = 0.0f;
        float sR = 0.0f, sG = 0.0f, sB = 0.0f, sC = 0.0f;
        // K samples per pass: the specular lobe's rare branches (clearcoat half vector, uniform-slope fallback) of the K
        // samples are evaluated packed (slow_push / slow_run / slow_pop above); each lobe's sums still grow in sample order
        for (int s0 = sub; s0 - sub < a.spp; s0 += K * G) {      // the same trip count in every lane
            int cnt = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const bool ok = s < a.spp;
                const int sc = ok ? s : 0;
                disney_spec_push<K>(slow, k, cnt, ok, d, w, bits_u01(tab[0][sc] ^ sx), bits_u01(tab[1][sc] ^ sy));
            }
            slow_run<K>(slow, cnt);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                float td[3] = { 0.0f, 0.0f, 0.0f }, ts[3] = { 0.0f, 0.0f, 0.0f };
                if (s < a.spp) {
                    // diffuse lobe (setSampleType(AI_RAY_DIFFUSE), src/rlDisney.cpp:242)
                    {
                        float rx = bits_u01(tab[0][s] ^ dx), ry = bits_u01(tab[1][s] ^ dy);
                        V3 L = cosine_hemisphere(d.fr, rx, ry);
                        float r, g, b, pdf;
                        disney_eval_pdf<true, true, true>(d, L, r, g, b, pdf);
                        if (pdf > kEps) { td[0] = r / pdf; td[1] = g / pdf; td[2] = b / pdf; dC += 1.0f; }
                        if (a.streamed && live) {
                            int64_t o = (int64_t)s * a.n + i;
                            st3(a.st.wi, o, L); strgb(a.st.f, o, r, g, b); stg(a.st.pdf, o, pdf);
                        }
                    }
                    // specular lobe (setSampleType(AI_RAY_GLOSSY), src/rlDisney.cpp:281,289)
                    {
                        V3 L = disney_spec_pop<K>(slow, k, d, w);
                        float r, g, b, pdf;
                        disney_eval_pdf<false, true, true>(d, L, r, g, b, pdf);
                        if (pdf > kEps) { ts[0] = r / pdf; ts[1] = g / pdf; ts[2] = b / pdf; sC += 1.0f; }   // :309
                        if (a.streamed && live) {
                            int64_t o = ((int64_t)a.spp + s) * a.n + i;
                            st3(a.st.wi, o, L); strgb(a.st.f, o, r, g, b); stg(a.st.pdf, o, pdf);
                        }
                    }
                }
                fold<G>(dR, td[0]); fold<G>(dG, td[1]); fold<G>(dB, td[2]);
                fold<G>(sR, ts[0]); fold<G>(sG, ts[1]); fold<G>(sB, ts[2]);
            }
        }
        if (G > 1) { dC = group_sum<G>(dC); sC = group_sum<G>(sC); }      // counts: integers, any order
        if (live && sub == 0) {
            const DisneyIntIO ao = RLS_INT_ARGS(a);
            strgb(ao.dsum, i, dR, dG, dB); stg(ao.dcount, i, dC);
            strgb(ao.ssum, i, sR, sG, sB); stg(ao.scount, i, sC);
        }
    }
}

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void disney_integrate_kernel(DisneyIntIO a)
{
    disney_integrate_body<G, FAST_MATH>(a);
}

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void disney_integrate_kernel_stamped(DisneyIntIO a, unsigned long long *stamps)
{
    ClockStamp<1> cs;
    cs.begin();
    disney_integrate_body<G, FAST_MATH>(a);
    cs.end(stamps);
}

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

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void sss_scatter_kernel(ScatterIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    stage_libm_tables();
    stage_table(tab, a.spp);
    const SceneRegs sc = scene_regs(a.scene);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const rls_sss_closure &c = a.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        NdProfile p = scatter_profile(c, pk);
        Frame fr = sss_frame(ld3(c.N, ii), ld3(c.T, ii), c.has_dPdu != 0);
        const V3 Po = ld3(a.P, ii);
        float br, bg, bb;
        ldrgb(c.sss_color, pk, br, bg, bb);
        const uint32_t sx = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream);
        const uint32_t sy = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + 1);

        float accR, accG, accB, accD;
        scatter_loop<G>(p, fr, Po, sc, tab, a.spp, sub, sx, sy, accR, accG, accB, accD);
        if (live && sub == 0) {
            const float inv = 1.0f / (float)a.spp;                               // AiSamplerGetSampleInvCount
            strgb(a.result, i, br * accR * inv, bg * accG * inv, bb * accB * inv);
            if (a.depth) stg(a.depth, i, accD * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rlSkin's shader_evaluate over spp_n^2 samples per layer (src/rlSkin.cpp:174-246): per GGX lobe integrateGlossy's
// sample loop, whose evalSample calls build the mean Fresnel that getAvgReflectWeight (src/rlGgx.h:181-184) hands to
// the next layer -- sheenFresnel = avg * sheen_weight (:204), specular *= specular_weight * (1 - sheenFresnel) (:231),
// specularFresnel (:228), sssWeight *= 1 - specularFresnel * (1 - sheenFresnel) (:238) -- then integrateScatter *
// sssWeight (:244-246).  AiBRDFIntegrate is closed: its stand-in is the mean of eval/pdf over the samples under a
// uniform environment of radiance `env` (parity unpinned); the light loops of :193-198,217-222 -> ggx_light_loops.
using rlsh::SkinIntIO;

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void skin_integrate_kernel(SkinIntIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<1> slow;                   // the lobes run the plain loops here (ggx_glossy_loop, PACK = false)
    stage_libm_tables();
    stage_table(tab, a.spp);
    const SceneRegs sc = scene_regs(a.scene);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    const float inv = 1.0f / (float)a.spp;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const rls_skin_closure &c = a.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        const V3 wo = ld3(c.wo, ii), N = ld3(c.N, ii), T = ld3(c.T, ii);
        uint32_t scr[6];
#pragma unroll
        for (int k = 0; k < 6; k++) scr[k] = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + k);
        Frame gfr;
        gfr.N = N; gfr.U = T; gfr.V = cross(N, T);
        const V3 local = vndf_local(wo, gfr);       // shared by the two lobes (same frame, same view)
        const V3 P = ld3(a.P, ii);

        float sheenFresnel = 0.0f, specularFresnel = 0.0f;
        float shR = 0.0f, shG = 0.0f, shB = 0.0f, spR = 0.0f, spG = 0.0f, spB = 0.0f;
        const float sheenWeight = ldp(c.sheen_weight, pk);
        // the group takes the branch together: the weights are per point, the G lanes of a group share the point
        if (sheenWeight > kEps) {                                                     // :191
            float cr, cg, cb;
            ldrgb(c.sheen_color, pk, cr, cg, cb);
            Ggx g = ggx_make<true>(wo, N, T, false, cr, cg, cb, ldp(c.sheen_ior, pk), ldp(c.sheen_roughness, pk), 0.0f);
            VndfView w = vndf_view_from(local, g.ax, g.ay);
            float lit[3], lf, lc, aF;
            ggx_light_loops<G>(g, w, N, P, a, tab, a.spp, sub, inv, a.seed, a.first + (uint64_t)ii, 3,
                               lit, lf, lc);                                          // :193-198
            ggx_glossy_loop<G, 1, false>(slow, g, w, tab, a.spp, sub, scr[0], scr[1], shR, shG, shB, aF, lf);
            // integrateGlossy returns black for a small colour without sampling (src/rlGgx.h:174-176); the light
            // loop samples regardless; getAvgReflectWeight (181-184) = sum / count over both, 1 when none were drawn
            const bool small = absf(cr) < kEps && absf(cg) < kEps && absf(cb) < kEps;
            const float fsum = small ? lf : aF, fcnt = small ? lc : lc + (float)a.spp;
            const float avg = fcnt > 0.0f ? R_DIV(fsum, fcnt) : 1.0f;
            if (small) { shR = 0.0f; shG = 0.0f; shB = 0.0f; }
            sheenFresnel = avg * sheenWeight;                                         // :204
            shR = lit[0] + shR * inv * a.env[0]; shG = lit[1] + shG * inv * a.env[1]; shB = lit[2] + shB * inv * a.env[2];
        }
        shR *= sheenWeight; shG *= sheenWeight; shB *= sheenWeight;                   // :207
        const float specWeight = ldp(c.specular_weight, pk);
        if (specWeight > kEps) {                                                      // :214
            float cr, cg, cb;
            ldrgb(c.specular_color, pk, cr, cg, cb);
            Ggx g = ggx_make<true>(wo, N, T, false, cr, cg, cb, ldp(c.specular_ior, pk), ldp(c.specular_roughness, pk), 0.0f);
            VndfView w = vndf_view_from(local, g.ax, g.ay);
            float lit[3], lf, lc, aF;
            ggx_light_loops<G>(g, w, N, P, a, tab, a.spp, sub, inv, a.seed, a.first + (uint64_t)ii, 5,
                               lit, lf, lc);                                          // :217-222
            ggx_glossy_loop<G, 1, false>(slow, g, w, tab, a.spp, sub, scr[2], scr[3], spR, spG, spB, aF, lf);
            const bool small = absf(cr) < kEps && absf(cg) < kEps && absf(cb) < kEps;
            const float fsum = small ? lf : aF, fcnt = small ? lc : lc + (float)a.spp;
            const float avg = fcnt > 0.0f ? R_DIV(fsum, fcnt) : 1.0f;
            if (small) { spR = 0.0f; spG = 0.0f; spB = 0.0f; }
            specularFresnel = avg * specWeight;                                       // :228
            spR = lit[0] + spR * inv * a.env[0]; spG = lit[1] + spG * inv * a.env[1]; spB = lit[2] + spB * inv * a.env[2];
        }
        const float sw = specWeight * (1.0f - sheenFresnel);                          // :231
        spR *= sw; spG *= sw; spB *= sw;

        const float mult = ldp(c.sss_dist_multiplier, pk);                            // :235-236
        float sssWeight = ldp(c.sss_weight, pk);
        sssWeight *= 1.0f - specularFresnel * (1.0f - sheenFresnel);                  // :238
        float ssR = 0.0f, ssG = 0.0f, ssB = 0.0f;
        if (!(sssWeight < kEps)) {                                                    // :244
            NdProfile p = nd_make<true>(ldp(c.sss_scatter_dist[0], pk) * mult, ldp(c.sss_scatter_dist[1], pk) * mult,
                                  ldp(c.sss_scatter_dist[2], pk) * mult);
            Frame fr = sss_frame(N, T, true);
            float br, bg, bb, accD;
            ldrgb(c.sss_color, pk, br, bg, bb);
            scatter_loop<G>(p, fr, P, sc, tab, a.spp, sub, scr[4], scr[5], ssR, ssG, ssB, accD);
            ssR = br * ssR * inv * sssWeight; ssG = bg * ssG * inv * sssWeight; ssB = bb * ssB * inv * sssWeight;
        }
        if (live && sub == 0) {
            strgb(a.sheen, i, shR, shG, shB);
            strgb(a.specular, i, spR, spG, spB);
            strgb(a.sss, i, ssR, ssG, ssB);
            if (a.out.r) strgb(a.out, i, shR + spR + ssR, shG + spG + ssG, shB + spB + ssB);   // sg->out.RGB, :254
            if (a.sheenFresnel) stg(a.sheenFresnel, i, sheenFresnel);
            if (a.specularFresnel) stg(a.specularFresnel, i, specularFresnel);
            if (a.sssWeight) stg(a.sssWeight, i, sssWeight);
        }
    }
}

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

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void ggx_refract_integrate_kernel(RefractIntIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<RLS_SPEC_BLOCK> slow;
    stage_libm_tables();   // atanf range table (+ expf / logf / powf tables) -> LDS
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const rls_ggx_closure &c = a.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        V3 wo = ld3(c.wo, ii), N = ld3(c.N, ii), T = ld3(c.T, ii);
        float kr, kg, kb;
        ldrgb(c.KsColor, pk, kr, kg, kb);
        bool exiting = c.exiting ? (c.exiting[ii] != 0) : false;
        Ggx g = ggx_make(wo, N, T, exiting, kr, kg, kb, ldp(c.ior, pk), ldp(c.specularRoughness, pk),
                         ldp(c.anisotropic, pk));
        float acc, tir;
        if (a.traced) {
            VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
            const uint32_t sx = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream);
            const uint32_t sy = hash_u32(a.seed, a.first + (uint64_t)ii, kScrambleStream + 1);
            ggx_refract_loop<G>(slow, g, w, tab, a.spp, sub, sx, sy, acc, tir);
        } else {
            ggx_refract_untraced(g, acc, tir);
        }
        if (live && sub == 0) {
            strgb(a.result, i, a.env[0] * acc, a.env[1] * acc, a.env[2] * acc);
            if (a.tir) stg(a.tir, i, tir);
        }
    }
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

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void ggx_direct_kernel(LightIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<RLS_SPEC_BLOCK> slow;
    stage_libm_tables();   // atanf range table (+ expf / logf / powf tables) -> LDS
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    const float inv = 1.0f / (float)a.spp;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const rls_ggx_closure &c = a.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        V3 wo = ld3(c.wo, ii), N = ld3(c.N, ii), T = ld3(c.T, ii);
        float kr, kg, kb;
        ldrgb(c.KsColor, pk, kr, kg, kb);
        bool exiting = c.exiting ? (c.exiting[ii] != 0) : false;
        Ggx g = ggx_make(wo, N, T, exiting, kr, kg, kb, ldp(c.ior, pk), ldp(c.specularRoughness, pk),
                         ldp(c.anisotropic, pk));
        VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
        OrenNayar on = oren_nayar_make(N, ldp(a.sh.diffuseRoughness, pk));
        const float ks = ldp(a.sh.Ks, pk), kd = ldp(a.sh.Kd, pk);
        float dr, dg, db;
        ldrgb(a.sh.KdColor, pk, dr, dg, db);
        dr *= kd; dg *= kd; db *= kd;                                       // diffuseColor, src/rlGgx.cpp:279
        float oD[3], oS[3];
        ggx_direct_loops<G>(slow, g, w, on, wo, N, ld3(a.P, ii), !color_is_small(dr, dg, db), a, tab, a.spp, sub,
                            inv, a.seed, a.first + (uint64_t)ii, oD, oS);
        if (live && sub == 0) {
            strgb(a.ds, i, oS[0] * ks, oS[1] * ks, oS[2] * ks);            // specular *= specularWeight, :305
            strgb(a.dd, i, oD[0] * dr, oD[1] * dg, oD[2] * db);            // diffuse *= diffuseColor, :304
        }
    }
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

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_DISNEY_LIGHT_ATTR void disney_direct_kernel(DisneyLightIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<RLS_SPEC_BLOCK> slow;
    stage_libm_tables();
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    const float inv = 1.0f / (float)a.spp;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        RLS_DISNEY_LOAD(d, a.c, ii)
        VndfView w = vndf_view(d.view, d.fr, d.ax, d.ay);
        float oD[3], oS[3];
        disney_direct_loops<G>(slow, d, w, d.fr.N, ld3(a.P, ii), a, tab, a.spp, sub, inv, a.seed,
                               a.first + (uint64_t)ii, oD, oS);
        if (live && sub == 0) {
            strgb(a.dd, i, oD[0], oD[1], oD[2]);
            strgb(a.ds, i, oS[0], oS[1], oS[2]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// shader_evaluate of rlGgx and of rlDisney for a camera ray, whole: the loops above run back to back on one closure
// set-up (include/rlshaders_amd.h, rls_ggx_shade / rls_disney_shade).  Sample streams: light l 3 l .. 3 l + 2 (as in
// the light-loop entry points), then 24, 25, 26 for the indirect loops.
using rlsh::GgxShadeIO;
using rlsh::DisneyShadeIO;
constexpr uint32_t kShadeStream = 3 * RLS_MAX_LIGHTS;       // first sample stream after the lights'

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void ggx_shade_kernel(GgxShadeIO a)
{
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<RLS_SPEC_BLOCK> slow;
    stage_libm_tables();
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    const float inv = 1.0f / (float)a.spp;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const uint64_t idx = a.first + (uint64_t)ii;
        const rls_ggx_closure &c = a.c;
        const PIndex<int64_t> pk = pindex(c.materials, ii);      // parameters by reference (rls_material_index)
        V3 wo = ld3(c.wo, ii), N = ld3(c.N, ii), T = ld3(c.T, ii);
        float kr, kg, kb;
        ldrgb(c.KsColor, pk, kr, kg, kb);
        bool exiting = c.exiting ? (c.exiting[ii] != 0) : false;
        Ggx g = ggx_make(wo, N, T, exiting, kr, kg, kb, ldp(c.ior, pk), ldp(c.specularRoughness, pk),
                         ldp(c.anisotropic, pk));
        VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
        OrenNayar on = oren_nayar_make(N, ldp(a.sh.diffuseRoughness, pk));
        const float ks = ldp(a.sh.Ks, pk), kd = ldp(a.sh.Kd, pk), kt = ldp(a.sh.Kt, pk);
        float dr, dg, db, tr, tg, tb;
        ldrgb(a.sh.KdColor, pk, dr, dg, db);
        ldrgb(a.sh.KtColor, pk, tr, tg, tb);
        dr *= kd; dg *= kd; db *= kd;                                        // diffuseColor, src/rlGgx.cpp:279
        tr *= kt; tg *= kt; tb *= kt;                                        // ktColor, :308
        const bool sampleDiffuse = !color_is_small(dr, dg, db);              // :280 (Rr_diff = 0)
        // the light loop, :285-305
        float dD[3], dS[3];
        ggx_direct_loops<G>(slow, g, w, on, wo, N, ld3(a.P, ii), sampleDiffuse, a, tab, a.spp, sub, inv, a.seed,
                            idx, dD, dS);
        dD[0] *= dr; dD[1] *= dg; dD[2] *= db;
        dS[0] *= ks; dS[1] *= ks; dS[2] *= ks;
        // transmission, :307-309
        float tx[3] = { 0.0f, 0.0f, 0.0f };
        if (!color_is_small(tr, tg, tb)) {
            float acc, tir;
            if (a.traced) {
                ggx_refract_loop<G>(slow, g, w, tab, a.spp, sub, hash_u32(a.seed, idx, kScrambleStream + 2 * (kShadeStream + 1)),
                                    hash_u32(a.seed, idx, kScrambleStream + 2 * (kShadeStream + 1) + 1), acc, tir);
            } else {
                ggx_refract_untraced(g, acc, tir);
            }
            tx[0] = a.env[0] * acc * tr; tx[1] = a.env[1] * acc * tg; tx[2] = a.env[2] * acc * tb;
        }
        // indirect diffuse, :315-319: AiBRDFIntegrate over the Oren-Nayar closure -> mean of brdf / pdf x env
        float iD[3] = { 0.0f, 0.0f, 0.0f };
        if (sampleDiffuse) {
            const uint32_t sx = hash_u32(a.seed, idx, kScrambleStream + 2 * (kShadeStream + 2));
            const uint32_t sy = hash_u32(a.seed, idx, kScrambleStream + 2 * (kShadeStream + 2) + 1);
            float acc = 0.0f;
            for (int s0 = 0; s0 < a.spp; s0 += G) {
                const int s = s0 + sub;
                float t = 0.0f;
                if (s < a.spp) {
                    V3 Ld = cosine_hemisphere(g.fr, bits_u01(tab[0][s] ^ sx), bits_u01(tab[1][s] ^ sy));
                    float pd = oren_nayar_pdf(on, Ld);
                    if (pd > 0.0f) t = R_DIV(oren_nayar_brdf(on, wo, Ld), pd);
                }
                fold<G>(acc, t);
            }
            acc *= inv;
            iD[0] = dr * (acc * a.env[0]); iD[1] = dg * (acc * a.env[1]); iD[2] = db * (acc * a.env[2]);
        }
        // indirect glossy, :321: integrateGlossy (black for a small colour, src/rlGgx.h:174-176) x specularWeight
        float iS[3] = { 0.0f, 0.0f, 0.0f };
        if (!color_is_small(kr, kg, kb)) {
            float aR, aG, aB, aF;
            ggx_glossy_loop<G>(slow, g, w, tab, a.spp, sub, hash_u32(a.seed, idx, kScrambleStream + 2 * kShadeStream),
                               hash_u32(a.seed, idx, kScrambleStream + 2 * kShadeStream + 1), aR, aG, aB, aF);
            iS[0] = aR * inv * a.env[0] * ks; iS[1] = aG * inv * a.env[1] * ks; iS[2] = aB * inv * a.env[2] * ks;
        }
        if (live && sub == 0) {
            strgb(a.dd, i, dD[0], dD[1], dD[2]);
            strgb(a.ds, i, dS[0], dS[1], dS[2]);
            strgb(a.refr, i, tx[0], tx[1], tx[2]);
            strgb(a.id, i, iD[0], iD[1], iD[2]);
            strgb(a.is, i, iS[0], iS[1], iS[2]);
            // result = diffuse + specular + transmission (:311); result += indirectDiffuse + indirectGlossy (:323)
            if (a.out.r) strgb(a.out, i, ((dD[0] + dS[0]) + tx[0]) + (iD[0] + iS[0]), ((dD[1] + dS[1]) + tx[1]) + (iD[1] + iS[1]),
                               ((dD[2] + dS[2]) + tx[2]) + (iD[2] + iS[2]));
        }
    }
}

template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_DISNEY_LIGHT_ATTR void disney_shade_kernel(DisneyShadeIO a)
{
    constexpr int K = RLS_SPEC_BLOCK;
    __shared__ uint32_t tab[2][kMaxSpp];
    __shared__ SlowLds<K> slow;
    stage_libm_tables();
    stage_table(tab, a.spp);
    const int sub = threadIdx.x % G;
    const int64_t groups_per_block = rlsh::kBlock / G;
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t rounds = (a.n + stride - 1) / stride;
    const float inv = 1.0f / (float)a.spp;
    int64_t i = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    for (int64_t it = 0; it < rounds; it++, i += stride) {
        const bool live = i < a.n;
        const int64_t ii = live ? i : a.n - 1;
        const uint64_t idx = a.first + (uint64_t)ii;
        RLS_DISNEY_LOAD(d, a.c, ii)
        VndfView w = vndf_view(d.view, d.fr, d.ax, d.ay);
        // the light loop, src/rlDisney.cpp:695-705
        float dD[3], dS[3];
        disney_direct_loops<G>(slow, d, w, d.fr.N, ld3(a.P, ii), a, tab, a.spp, sub, inv, a.seed, idx, dD, dS);
        // integrateDiffuse / integrateGlossy (:718-719, 240-243, 279-283): AiBRDFIntegrate over the triple -> the sum of
        // brdf / pdf over the valid samples (:309) x AiSamplerGetSampleInvCount x env
        uint32_t scr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) scr[k] = hash_u32(a.seed, idx, kScrambleStream + 2 * kShadeStream + k);
        float iR = 0.0f, iG = 0.0f, iB = 0.0f, gR = 0.0f, gG = 0.0f, gB = 0.0f;
        for (int s0 = sub; s0 - sub < a.spp; s0 += K * G) {      // K samples per pass, as disney_integrate_kernel
            int cnt = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                const int sc = s < a.spp ? s : 0;
                disney_spec_push<K>(slow, k, cnt, s < a.spp, d, w, bits_u01(tab[0][sc] ^ scr[2]), bits_u01(tab[1][sc] ^ scr[3]));
            }
            slow_run<K>(slow, cnt);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int s = s0 + k * G;
                float td[3] = { 0.0f, 0.0f, 0.0f }, ts[3] = { 0.0f, 0.0f, 0.0f };
                if (s < a.spp) {
                    {
                        V3 L = cosine_hemisphere(d.fr, bits_u01(tab[0][s] ^ scr[0]), bits_u01(tab[1][s] ^ scr[1]));
                        float r, g, b, pdf;
                        disney_eval_pdf<true, true, true>(d, L, r, g, b, pdf);
                        if (pdf > kEps) { td[0] = r / pdf; td[1] = g / pdf; td[2] = b / pdf; }
                    }
                    {
                        V3 L = disney_spec_pop<K>(slow, k, d, w);
                        float r, g, b, pdf;
                        disney_eval_pdf<false, true, true>(d, L, r, g, b, pdf);
                        if (pdf > kEps) { ts[0] = r / pdf; ts[1] = g / pdf; ts[2] = b / pdf; }
                    }
                }
                fold<G>(iR, td[0]); fold<G>(iG, td[1]); fold<G>(iB, td[2]);
                fold<G>(gR, ts[0]); fold<G>(gG, ts[1]); fold<G>(gB, ts[2]);
            }
        }
        const float iD[3] = { iR * inv * a.env[0], iG * inv * a.env[1], iB * inv * a.env[2] };
        const float iS[3] = { gR * inv * a.env[0], gG * inv * a.env[1], gB * inv * a.env[2] };
        if (live && sub == 0) {
            strgb(a.dd, i, dD[0], dD[1], dD[2]);
            strgb(a.ds, i, dS[0], dS[1], dS[2]);
            strgb(a.id, i, iD[0], iD[1], iD[2]);
            strgb(a.is, i, iS[0], iS[1], iS[2]);
            // result = diffuse + specular (:712); result += indirectDiffuse + indirectGlossy (:722)
            if (a.out.r) strgb(a.out, i, (dD[0] + dS[0]) + (iD[0] + iS[0]), (dD[1] + dS[1]) + (iD[1] + iS[1]),
                               (dD[2] + dS[2]) + (iD[2] + iS[2]));
        }
    }
}

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

// BASELINE config 3 (one lane per point) under rls_diag_clock_stamps_begin: the stamped instantiation
inline rls_status launch_disney_stamped(rls_context *ctx, const rlsh::DisneyIntIO &io, const char *name)
{
    hipLaunchKernelGGL(disney_integrate_kernel_stamped<1>, rlsh::grid_for(ctx, io.n, rlsh::kBlock), dim3(rlsh::kBlock), 0, ctx->stream,
                       io, ctx->stamps);
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

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_ggx_integrate(rls_context *ctx, int g, const rlsh::GgxIntIO *io)
{
    return launch_g(ctx, ggx_integrate_kernel<1>, ggx_integrate_kernel<4>, ggx_integrate_kernel<16>,
                    ggx_integrate_kernel<64>, g, *io, "rls_ggx_integrate[fast]");
}
RLS_HIDDEN rls_status rls_fast_disney_integrate(rls_context *ctx, int g, const rlsh::DisneyIntIO *io)
{
    if (ctx->stamps && g == 1) return launch_disney_stamped(ctx, *io, "rls_disney_integrate[fast, stamped]");
    return launch_g(ctx, disney_integrate_kernel<1>, disney_integrate_kernel<4>, disney_integrate_kernel<16>,
                    disney_integrate_kernel<64>, g, *io, "rls_disney_integrate[fast]");
}
RLS_HIDDEN rls_status rls_fast_ggx_direct(rls_context *ctx, int g, const rlsh::LightIO *io)
{
    return launch_g(ctx, ggx_direct_kernel<1>, ggx_direct_kernel<4>, ggx_direct_kernel<16>,
                    ggx_direct_kernel<64>, g, *io, "rls_ggx_direct_lighting[fast]");
}
RLS_HIDDEN rls_status rls_fast_ggx_shade(rls_context *ctx, int g, const rlsh::GgxShadeIO *io)
{
    return launch_g(ctx, ggx_shade_kernel<1>, ggx_shade_kernel<4>, ggx_shade_kernel<16>, ggx_shade_kernel<64>, g, *io,
                    "rls_ggx_shade[fast]");
}
RLS_HIDDEN rls_status rls_fast_disney_shade(rls_context *ctx, int g, const rlsh::DisneyShadeIO *io)
{
    return launch_g(ctx, disney_shade_kernel<1>, disney_shade_kernel<4>, disney_shade_kernel<16>, disney_shade_kernel<64>, g,
                    *io, "rls_disney_shade[fast]");
}
RLS_HIDDEN rls_status rls_fast_disney_direct(rls_context *ctx, int g, const rlsh::DisneyLightIO *io)
{
    return launch_g(ctx, disney_direct_kernel<1>, disney_direct_kernel<4>, disney_direct_kernel<16>,
                    disney_direct_kernel<64>, g, *io, "rls_disney_direct_lighting[fast]");
}
RLS_HIDDEN rls_status rls_fast_skin_integrate(rls_context *ctx, int g, const rlsh::SkinIntIO *io)
{
    return launch_g(ctx, skin_integrate_kernel<1>, skin_integrate_kernel<4>, skin_integrate_kernel<16>,
                    skin_integrate_kernel<64>, g, *io, "rls_skin_integrate[fast]");
}
RLS_HIDDEN rls_status rls_fast_ggx_refract_integrate(rls_context *ctx, int g, const rlsh::RefractIntIO *io)
{
    return launch_g(ctx, ggx_refract_integrate_kernel<1>, ggx_refract_integrate_kernel<4>, ggx_refract_integrate_kernel<16>,
                    ggx_refract_integrate_kernel<64>, g, *io, "rls_ggx_integrate_refract[fast]");
}
RLS_HIDDEN rls_status rls_fast_sss_scatter(rls_context *ctx, int g, const rlsh::ScatterIO *io)
{
    return launch_g(ctx, sss_scatter_kernel<1>, sss_scatter_kernel<4>, sss_scatter_kernel<16>,
                    sss_scatter_kernel<64>, g, *io, "rls_sss_integrate_scatter[fast]");
}
#else
RLS_HIDDEN rls_status rls_fast_ggx_integrate(rls_context *ctx, int g, const rlsh::GgxIntIO *io);
RLS_HIDDEN rls_status rls_fast_disney_integrate(rls_context *ctx, int g, const rlsh::DisneyIntIO *io);
RLS_HIDDEN rls_status rls_fast_sss_scatter(rls_context *ctx, int g, const rlsh::ScatterIO *io);
RLS_HIDDEN rls_status rls_fast_ggx_direct(rls_context *ctx, int g, const rlsh::LightIO *io);
RLS_HIDDEN rls_status rls_fast_skin_integrate(rls_context *ctx, int g, const rlsh::SkinIntIO *io);
RLS_HIDDEN rls_status rls_fast_disney_direct(rls_context *ctx, int g, const rlsh::DisneyLightIO *io);
RLS_HIDDEN rls_status rls_fast_ggx_shade(rls_context *ctx, int g, const rlsh::GgxShadeIO *io);
RLS_HIDDEN rls_status rls_fast_disney_shade(rls_context *ctx, int g, const rlsh::DisneyShadeIO *io);
RLS_HIDDEN rls_status rls_fast_ggx_refract_integrate(rls_context *ctx, int g, const rlsh::RefractIntIO *io);

extern "C" {

rls_status rls_skin_integrate(rls_context *ctx, int64_t n, const rls_skin_closure *c, rls_cvec3 P,
                              const rls_sss_scene *scene, const float env[3],
                              const rls_sphere_light *lights, int n_lights,
                              int spp_n, uint32_t seed, uint64_t first_index, const rls_skin_integrate_out *out)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && scene != nullptr && out != nullptr && env != nullptr, "closure, scene, env or out is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T) && rlsh::has3(P), "wo/N/T/P plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->sss_color) && rlsh::ok_rgb(c->specular_color) && rlsh::ok_rgb(c->sheen_color),
                "colour planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(scene->geometry == RLS_SCENE_PLANE || scene->geometry == RLS_SCENE_SPHERE, "unknown scene geometry");
    RLS_REQUIRE(rlsh::has3(out->sheen) && rlsh::has3(out->specular) && rlsh::has3(out->sss), "NULL AOV plane");
    RLS_REQUIRE(rlsh::has3(out->out) || (!out->out.r && !out->out.g && !out->out.b), "out planes must be all set or all NULL");
    rlsh::SkinIntIO io = {};
    if (rls_status st = copy_lights(lights, n_lights, 0, io.lights, &io.nl)) return st;
    io.c = *c; io.P = P; io.scene = *scene; io.env[0] = env[0]; io.env[1] = env[1]; io.env[2] = env[2];
    io.sheen = out->sheen; io.specular = out->specular; io.sss = out->sss; io.out = out->out;
    io.sheenFresnel = out->sheenFresnel; io.specularFresnel = out->specularFresnel; io.sssWeight = out->sssWeight;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_skin_integrate(ctx, g, &io);
    return launch_g(ctx, skin_integrate_kernel<1>, skin_integrate_kernel<4>, skin_integrate_kernel<16>,
                    skin_integrate_kernel<64>, g, io, "rls_skin_integrate");
}

rls_status rls_ggx_integrate_refract(rls_context *ctx, int64_t n, const rls_ggx_closure *c, int traced,
                                     const float env[3], int spp_n, uint32_t seed, uint64_t first_index,
                                     rls_rgb result, float *tir_fraction)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && env != nullptr, "closure or env is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->KsColor), "KsColor planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(result), "NULL output plane");
    rlsh::RefractIntIO io = {};
    io.c = *c; io.env[0] = env[0]; io.env[1] = env[1]; io.env[2] = env[2]; io.traced = traced ? 1 : 0;
    io.result = result; io.tir = tir_fraction;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = traced ? pick_group(ctx, n, io.spp) : 1;
    if (ctx->fast) return rls_fast_ggx_refract_integrate(ctx, g, &io);
    return launch_g(ctx, ggx_refract_integrate_kernel<1>, ggx_refract_integrate_kernel<4>, ggx_refract_integrate_kernel<16>,
                    ggx_refract_integrate_kernel<64>, g, io, "rls_ggx_integrate_refract");
}

rls_status rls_ggx_integrate(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                             int spp_n, uint32_t seed, uint64_t first_index,
                             rls_rgb sum_f_over_pdf, float *avg_reflect_weight)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->KsColor), "KsColor planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(sum_f_over_pdf) && avg_reflect_weight, "NULL output plane");
    GgxIntIO io = {};
    io.c = *c; io.sum = sum_f_over_pdf; io.avgF = avg_reflect_weight; io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_ggx_integrate(ctx, g, &io);
    return launch_g(ctx, ggx_integrate_kernel<1>, ggx_integrate_kernel<4>, ggx_integrate_kernel<16>,
                    ggx_integrate_kernel<64>, g, io, "rls_ggx_integrate");
}

rls_status rls_disney_integrate(rls_context *ctx, int64_t n, const rls_disney_closure *c,
                                int spp_n, uint32_t seed, uint64_t first_index,
                                rls_rgb diffuse_sum, float *diffuse_count,
                                rls_rgb specular_sum, float *specular_count,
                                const rls_disney_stream_out *stream)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->base_color), "base_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(diffuse_sum) && diffuse_count && rlsh::has3(specular_sum) && specular_count,
                "NULL output plane");
    DisneyIntIO io = {};
    io.c = *c; io.dsum = diffuse_sum; io.dcount = diffuse_count; io.ssum = specular_sum; io.scount = specular_count;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    if (stream) {
        RLS_REQUIRE(rlsh::has3(stream->wi) && rlsh::has3(stream->f) && stream->pdf, "NULL streamed-output plane");
        io.st = *stream;
        io.streamed = 1;
    }
    // streamed planes are sample-major: one lane per point keeps every store coalesced
    int g = io.streamed ? 1 : pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_disney_integrate(ctx, g, &io);
    if (ctx->stamps && g == 1) return launch_disney_stamped(ctx, io, "rls_disney_integrate[stamped]");
    return launch_g(ctx, disney_integrate_kernel<1>, disney_integrate_kernel<4>, disney_integrate_kernel<16>,
                    disney_integrate_kernel<64>, g, io, "rls_disney_integrate");
}

// Streamed mode in chunks of the point range (2^26 points x 128 triples x 28 B = 241 GB does not fit beside the
// inputs): each chunk is one launch over [p0, p0 + count) with every per-point plane pointer (by reference: the material
// ids instead of the parameter columns) advanced by p0 and the sampler's first_index by p0, so the samples are those of
// the unchunked call.
rls_status rls_disney_integrate_chunked(rls_context *ctx, int64_t n, const rls_disney_closure *c,
                                        int spp_n, uint32_t seed, uint64_t first_index,
                                        rls_rgb diffuse_sum, float *diffuse_count,
                                        rls_rgb specular_sum, float *specular_count,
                                        int64_t chunk_points, const rls_disney_stream_out *chunk,
                                        rls_disney_chunk_fn consume, void *user)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(chunk_points >= 1, "chunk_points < 1");
    RLS_REQUIRE(chunk != nullptr, "chunk buffers are NULL");
    if (ctx->capturing && consume != nullptr) {
        // the consumer runs on the host between chunks; a graph replay would drop it and lose every chunk but the last
        rlsh::set_error("rls_disney_integrate_chunked: a consumer callback cannot be recorded into a launch graph");
        return RLS_ERR_UNSUPPORTED;
    }
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    for (int64_t p0 = 0; p0 < n; p0 += chunk_points) {
        const int64_t count = n - p0 < chunk_points ? n - p0 : chunk_points;
        rls_disney_closure cc = *c;
        cc.wo = adv(c->wo, p0); cc.N = adv(c->N, p0); cc.T = adv(c->T, p0);
        if (c->materials.id) {
            // parameters by reference: the parameter pointers are per-MATERIAL columns of materials.count floats and stay
            // where they are; what belongs to the chunk's points is their material ids
            cc.materials.id = c->materials.id + p0;
        } else {
            cc.base_color = adv(c->base_color, p0);
            cc.subsurface = adv(c->subsurface, p0); cc.metallic = adv(c->metallic, p0); cc.specular = adv(c->specular, p0);
            cc.specular_tint = adv(c->specular_tint, p0); cc.roughness = adv(c->roughness, p0);
            cc.anisotropic = adv(c->anisotropic, p0); cc.sheen = adv(c->sheen, p0); cc.sheen_tint = adv(c->sheen_tint, p0);
            cc.clearcoat = adv(c->clearcoat, p0); cc.clearcoat_gloss = adv(c->clearcoat_gloss, p0);
        }
        rls_status st = rls_disney_integrate(ctx, count, &cc, spp_n, seed, first_index + (uint64_t)p0,
                                             adv(diffuse_sum, p0), diffuse_count ? diffuse_count + p0 : nullptr,
                                             adv(specular_sum, p0), specular_count ? specular_count + p0 : nullptr, chunk);
        if (st != RLS_OK) return st;
        if (consume) {
            int rc = consume(user, p0, count, chunk);
            if (rc != 0) {
                rlsh::set_error("rls_disney_integrate_chunked: consumer returned %d at point %lld", rc, (long long)p0);
                return RLS_ERR_ABORTED;
            }
        }
    }
    return RLS_OK;
}

rls_status rls_sss_integrate_scatter(rls_context *ctx, int64_t n, const rls_sss_closure *c, rls_cvec3 P,
                                     const rls_sss_scene *scene, int spp_n, uint32_t seed, uint64_t first_index,
                                     rls_rgb result, float *mean_depth)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && scene != nullptr, "closure or scene is NULL");
    RLS_REQUIRE(rlsh::has3(c->N) && rlsh::has3(c->T) && rlsh::has3(P), "N/T/P plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->sss_color), "sss_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(scene->geometry == RLS_SCENE_PLANE || scene->geometry == RLS_SCENE_SPHERE, "unknown scene geometry");
    RLS_REQUIRE(rlsh::has3(result), "NULL output plane");
    ScatterIO io = {};
    io.c = *c; io.P = P; io.scene = *scene; io.result = result; io.depth = mean_depth;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_sss_scatter(ctx, g, &io);
    return launch_g(ctx, sss_scatter_kernel<1>, sss_scatter_kernel<4>, sss_scatter_kernel<16>,
                    sss_scatter_kernel<64>, g, io, "rls_sss_integrate_scatter");
}

rls_status rls_ggx_direct_lighting(rls_context *ctx, int64_t n, const rls_ggx_closure *c, const rls_ggx_shader *sh,
                                   rls_cvec3 P, const rls_sphere_light *lights, int n_lights, int spp_n, uint32_t seed,
                                   uint64_t first_index, rls_rgb direct_diffuse, rls_rgb direct_specular)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && sh != nullptr, "closure or shader is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T) && rlsh::has3(P), "wo/N/T/P plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->KsColor) && rlsh::ok_rgb(sh->KdColor), "colour planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(direct_diffuse) && rlsh::has3(direct_specular), "NULL output plane");
    LightIO io = {};
    if (rls_status st = copy_lights(lights, n_lights, 1, io.lights, &io.nl)) return st;
    io.c = *c; io.sh = *sh; io.P = P; io.dd = direct_diffuse; io.ds = direct_specular;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_ggx_direct(ctx, g, &io);
    return launch_g(ctx, ggx_direct_kernel<1>, ggx_direct_kernel<4>, ggx_direct_kernel<16>,
                    ggx_direct_kernel<64>, g, io, "rls_ggx_direct_lighting");
}

rls_status rls_ggx_shade(rls_context *ctx, int64_t n, const rls_ggx_closure *c, const rls_ggx_shader *sh, rls_cvec3 P,
                         const rls_sphere_light *lights, int n_lights, const float env[3], int traced, int spp_n,
                         uint32_t seed, uint64_t first_index, const rls_ggx_shade_out *out)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && sh != nullptr && env != nullptr && out != nullptr, "closure, shader, env or out is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T) && rlsh::has3(P), "wo/N/T/P plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->KsColor) && rlsh::ok_rgb(sh->KdColor) && rlsh::ok_rgb(sh->KtColor),
                "colour planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(out->direct_diffuse) && rlsh::has3(out->direct_specular) && rlsh::has3(out->refraction) &&
                rlsh::has3(out->indirect_diffuse) && rlsh::has3(out->indirect_specular), "NULL AOV plane");
    RLS_REQUIRE(rlsh::has3(out->out) || (!out->out.r && !out->out.g && !out->out.b), "out planes must be all set or all NULL");
    rlsh::GgxShadeIO io = {};
    if (rls_status st = copy_lights(lights, n_lights, 0, io.lights, &io.nl)) return st;
    io.c = *c; io.sh = *sh; io.P = P; io.env[0] = env[0]; io.env[1] = env[1]; io.env[2] = env[2]; io.traced = traced ? 1 : 0;
    io.dd = out->direct_diffuse; io.ds = out->direct_specular; io.refr = out->refraction; io.id = out->indirect_diffuse;
    io.is = out->indirect_specular; io.out = out->out;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_ggx_shade(ctx, g, &io);
    return launch_g(ctx, ggx_shade_kernel<1>, ggx_shade_kernel<4>, ggx_shade_kernel<16>, ggx_shade_kernel<64>, g, io,
                    "rls_ggx_shade");
}

rls_status rls_disney_shade(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 P,
                            const rls_sphere_light *lights, int n_lights, const float env[3], int spp_n, uint32_t seed,
                            uint64_t first_index, const rls_disney_shade_out *out)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && env != nullptr && out != nullptr, "closure, env or out is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T) && rlsh::has3(P), "wo/N/T/P plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->base_color), "base_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(out->direct_diffuse) && rlsh::has3(out->direct_specular) && rlsh::has3(out->indirect_diffuse) &&
                rlsh::has3(out->indirect_specular), "NULL AOV plane");
    RLS_REQUIRE(rlsh::has3(out->out) || (!out->out.r && !out->out.g && !out->out.b), "out planes must be all set or all NULL");
    rlsh::DisneyShadeIO io = {};
    if (rls_status st = copy_lights(lights, n_lights, 0, io.lights, &io.nl)) return st;
    io.c = *c; io.P = P; io.env[0] = env[0]; io.env[1] = env[1]; io.env[2] = env[2];
    io.dd = out->direct_diffuse; io.ds = out->direct_specular; io.id = out->indirect_diffuse; io.is = out->indirect_specular;
    io.out = out->out;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_disney_shade(ctx, g, &io);
    return launch_g(ctx, disney_shade_kernel<1>, disney_shade_kernel<4>, disney_shade_kernel<16>, disney_shade_kernel<64>, g,
                    io, "rls_disney_shade");
}

rls_status rls_disney_direct_lighting(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 P,
                                      const rls_sphere_light *lights, int n_lights, int spp_n, uint32_t seed,
                                      uint64_t first_index, rls_rgb direct_diffuse, rls_rgb direct_specular)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(spp_n >= 1 && spp_n * spp_n <= kMaxSpp, "spp_n must be in [1, 16]");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T) && rlsh::has3(P), "wo/N/T/P plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->base_color), "base_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    RLS_REQUIRE(rlsh::has3(direct_diffuse) && rlsh::has3(direct_specular), "NULL output plane");
    DisneyLightIO io = {};
    if (rls_status st = copy_lights(lights, n_lights, 1, io.lights, &io.nl)) return st;
    io.c = *c; io.P = P; io.dd = direct_diffuse; io.ds = direct_specular;
    io.n = n; io.spp = spp_n * spp_n; io.seed = seed; io.first = first_index;
    int g = pick_group(ctx, n, io.spp);
    if (ctx->fast) return rls_fast_disney_direct(ctx, g, &io);
    return launch_g(ctx, disney_direct_kernel<1>, disney_direct_kernel<4>, disney_direct_kernel<16>,
                    disney_direct_kernel<64>, g, io, "rls_disney_direct_lighting");
}

} // extern "C"

#endif // !RLS_FAST
