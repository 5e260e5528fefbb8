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
//
// The loops themselves -- samplers, the packed rare branches (SlowLds), integrateGlossy / integrateRefract / the light loops /
// integrateScatter for one shading point -- are rls_loops.hpp; this unit holds the kernels and entry points of SURVEY.md 8(a)
// rows a16 / a17 and BASELINE config 3: rls_ggx_integrate, rls_ggx_integrate_refract, rls_disney_integrate(_chunked).
// lights.hip (the light loops), scatter.hip (integrateScatter) and shade.hip (the three nodes' whole shader_evaluate)
// hold the 8(f) rows built on the same loops.
#include "rls_loops.hpp"

namespace {

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

#if RLS_DIAGNOSTICS
template <int G, int FAST_MATH = RLS_FAST>
__global__ RLS_INT_ATTR void disney_integrate_kernel_stamped(DisneyIntIO a, unsigned long long *stamps)
{
    ClockStamp<1> cs;
    cs.begin();
    disney_integrate_body<G, FAST_MATH>(a);
    cs.end(stamps);
}
#endif

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

#if RLS_DIAGNOSTICS
// BASELINE config 3 (one lane per point) under rls_diag_clock_stamps_begin: the stamped instantiation
inline rls_status launch_disney_stamped(rls_context *ctx, const rlsh::DisneyIntIO &io, unsigned long long *stamps, const char *name)
{
    hipLaunchKernelGGL(disney_integrate_kernel_stamped<1>, rlsh::grid_for(ctx, io.n, rlsh::kBlock), dim3(rlsh::kBlock), 0, ctx->stream,
                       io, stamps);
    return rlsh::check_launch(name);
}
#endif

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_ggx_integrate(rls_context *ctx, int g, const rlsh::GgxIntIO *io)
{
    return launch_g(ctx, ggx_integrate_kernel<1>, ggx_integrate_kernel<4>, ggx_integrate_kernel<16>,
                    ggx_integrate_kernel<64>, g, *io, "rls_ggx_integrate[fast]");
}
RLS_HIDDEN rls_status rls_fast_disney_integrate(rls_context *ctx, int g, const rlsh::DisneyIntIO *io)
{
#if RLS_DIAGNOSTICS
    if (unsigned long long *stamps = g == 1 ? rlsh::stamps_for_launch(ctx) : nullptr)
        return launch_disney_stamped(ctx, *io, stamps, "rls_disney_integrate[fast, stamped]");
#endif
    return launch_g(ctx, disney_integrate_kernel<1>, disney_integrate_kernel<4>, disney_integrate_kernel<16>,
                    disney_integrate_kernel<64>, g, *io, "rls_disney_integrate[fast]");
}
RLS_HIDDEN rls_status rls_fast_ggx_refract_integrate(rls_context *ctx, int g, const rlsh::RefractIntIO *io)
{
    return launch_g(ctx, ggx_refract_integrate_kernel<1>, ggx_refract_integrate_kernel<4>, ggx_refract_integrate_kernel<16>,
                    ggx_refract_integrate_kernel<64>, g, *io, "rls_ggx_integrate_refract[fast]");
}
#else
RLS_HIDDEN rls_status rls_fast_ggx_integrate(rls_context *ctx, int g, const rlsh::GgxIntIO *io);
RLS_HIDDEN rls_status rls_fast_disney_integrate(rls_context *ctx, int g, const rlsh::DisneyIntIO *io);
RLS_HIDDEN rls_status rls_fast_ggx_refract_integrate(rls_context *ctx, int g, const rlsh::RefractIntIO *io);

extern "C" {

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
#if RLS_DIAGNOSTICS
    if (unsigned long long *stamps = g == 1 ? rlsh::stamps_for_launch(ctx) : nullptr)
        return launch_disney_stamped(ctx, io, stamps, "rls_disney_integrate[stamped]");
#endif
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

} // extern "C"

#endif // !RLS_FAST
