// ggx.hip -- rlGgx closure kernels (rls::GgxSamplerT<VNDFKernel>, src/rlGgx.h:92-373 and
// src/rlGgx.cpp:14-99 of the reference) and their C-ABI entry points.  gfx950, wave64, one
// shading point per lane, planar SoA streams, XCD-contiguous tiles strided over the batch.
//
// Roofline: HBM nominally (EXACT arithmetic makes the kernels VALU-issue-bound: DESIGN.md section 5).  Algorithmic bytes per point (all planes streamed): reflect triple 64 B in
// (wo3 N3 T3 Ks3 rough ior xi2) + 32 B out (wi3 f3 pdf F) = 96 B; reflect+refract 72 B in +
// 48 B out = 120 B; +4 B when `anisotropic` is a stream.
#include "rls_internal.hpp"

using namespace rlsd;

namespace {

using namespace rlsh;   // GgxOp, GgxIO

#define RLS_GGX_ARGS(a0) reload_args(a0)

// MODE (checked on the host): STREAMED_ALL every closure parameter is a per-point plane (no stream-or-uniform tests in the
// loop); UNIFORM_MATERIAL roughness, ior and anisotropic are single values for the batch (an Arnold parameter is a constant
// unless a texture is linked to it; specColor may be either -- it enters no arithmetic of the constructor) and the
// parameter-only half of the constructor (ggx_material: aspect, alpha_x, alpha_y, the reciprocal ior) runs once per thread
// ahead of the tile loop, its results kept in scalar registers; MIXED tests parameter by parameter
enum { MIXED = 0, STREAMED_ALL = 1, UNIFORM_MATERIAL = 2 };

template <int MODE>
__device__ __forceinline__ Ggx load_closure(const rls_ggx_closure &c, Idx i, const GgxMaterial &um)
{
    constexpr bool STREAMED = MODE == STREAMED_ALL;
    V3 wo = ld3(c.wo, i), N = ld3(c.N, i), T = ld3(c.T, i);
    const PIndex<Idx> k = pindex<MODE == MIXED>(c.materials, i);       // parameters by reference run the MIXED kernel
    float kr, kg, kb;
    ldrgb<STREAMED>(c.KsColor, k, kr, kg, kb);
    if (MODE == UNIFORM_MATERIAL)
        return ggx_from_material(um, wo, N, T, c.exiting ? (c.exiting[i.full()] != 0) : false, kr, kg, kb);
    float rough = ldp<STREAMED>(c.specularRoughness, k);
    float ior = ldp<STREAMED>(c.ior, k);
    float aniso = ldp<STREAMED>(c.anisotropic, k);
    bool exiting = c.exiting ? (c.exiting[i.full()] != 0) : false;
    return ggx_make(wo, N, T, exiting, kr, kg, kb, ior, rough, aniso);
}

// FAST_MATH only tags the kernel name (profiles tell the two builds apart)
// occupancy per verb (waves per SIMD the register allocator must allow; rls_internal.hpp has the first measurement, at six).
// Re-measured at the end of round 3 (tools/ab.sh, two interleaved repetitions, 5 / 6 / 7 / 8 waves): reflect + refract
// 2.071 / 2.026 / 2.008 / 2.022 ... 2.054 / - / 2.032 / 2.023 ms on two boxes, the reflect triple 1.690 / 1.642 / 1.628, evalPdf alone
// 0.701 (6) / 0.674 (7) / 0.652 (8), evalBrdf alone 0.938 (6) / 0.953 (7) / 0.958 (8): eight waves (64 registers, 26 scalar
// registers spilled to lanes, no scratch) for everything but evalBrdf.
#ifndef RLS_GGX_WAVES
#define RLS_GGX_WAVES(OP) ((OP) == OP_EVAL ? 6 : 8)
#endif
#define RLS_GGX_ATTR(OP) __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_GGX_WAVES(OP), RLS_GGX_WAVES(OP))))
// the kernel body: inlined into ggx_kernel (the product) and into ggx_kernel_stamped (the diagnostic instantiation that
// brackets it with clock stamps, rls_internal.hpp ClockStamp).  a0 is the kernel's first parameter (reload_args).
template <int OP, int FAST_MATH, int MODE>
__device__ __forceinline__ void ggx_body(const GgxIO &a0)
{
    if (OP == OP_SAMPLE || OP == OP_FUSED || OP == OP_REFLECT_REFRACT || OP == OP_REFRACT || OP == OP_MICROFACET)
        stage_libm_tables();   // the range table of atanf -> LDS (visible-normal sampling calls atan2f twice)
    GgxMaterial um = {};
    if (MODE == UNIFORM_MATERIAL) {
        const rls_ggx_closure &c = a0.c;
        um = ggx_material_wave_uniform(ggx_material(c.ior.u, c.specularRoughness.u, c.anisotropic.u));
    }
    const TileRange tiles = tile_range(a0.n);
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const Idx i = make_idx(base);
        if (i.full() >= a0.n) continue;
        // plane pointers re-read from the kernarg segment where they are used (rls_internal.hpp, reload_args)
        const GgxIO a = RLS_GGX_ARGS(a0);
        Ggx g = load_closure<MODE>(a.c, i, um);

        if (OP == OP_SAMPLE || OP == OP_FUSED || OP == OP_REFLECT_REFRACT) {
            float rx = ldg(a.rx, i), ry = ldg(a.ry, i);
            VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
            // evalSample: src/rlGgx.h:97-107
            V3 M, M2;
            if (OP == OP_REFLECT_REFRACT) {
                // second sample on the same closure: the view analysis is reused, the uniform fallback shared
                float rx2 = ldg(a.rx2, i), ry2 = ldg(a.ry2, i);
                vndf_microfacet_pair(w, g.fr, rx, ry, w, g.fr, rx2, ry2, M, M2);
            } else {
                M = vndf_microfacet(w, g.fr, rx, ry);
            }
            V3 L = reflect_direction(g.view, M);
            float F = ggx_fresnel(g, L, M);
            {
                const GgxIO b = RLS_GGX_ARGS(a0);
                st3(b.wi, i, L);
                if (b.fresnel) stg(b.fresnel, i, F);
            }
            if (OP != OP_SAMPLE) {
                float fr, fg, fb, pdf;
                ggx_eval_pdf<true, true>(g, L, fr, fg, fb, pdf);
                const GgxIO b = RLS_GGX_ARGS(a0);
                strgb(b.f, i, fr, fg, fb);
                stg(b.pdf, i, pdf);
            }
            if (OP == OP_REFLECT_REFRACT) {
                V3 dir;
                ggx_refract(g, M2, dir);
                const float wgt = ggx_sample_weight(g, g.view, dir, M2);
                const GgxIO b = RLS_GGX_ARGS(a0);
                st3(b.wt, i, dir);
                stg(b.weight, i, wgt);
            }
        } else if (OP == OP_EVAL) {
            float fr, fg, fb;
            ggx_eval(g, ld3(a.cwi, i), fr, fg, fb);
            strgb(a.f, i, fr, fg, fb);
        } else if (OP == OP_PDF) {
            stg(a.pdf, i, ggx_pdf(g, ld3(a.cwi, i)));
        } else if (OP == OP_REFRACT) {
            float rx = ldg(a.rx, i), ry = ldg(a.ry, i);
            VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
            V3 M = vndf_microfacet(w, g.fr, rx, ry);
            V3 dir;
            bool ok = ggx_refract(g, M, dir);
            st3(a.wt, i, dir);
            stg(a.weight, i, ggx_sample_weight(g, g.view, dir, M));
            if (a.refracted) a.refracted[i.full()] = ok ? 1 : 0;
        } else if (OP == OP_MICROFACET) {
            float rx = ldg(a.rx, i), ry = ldg(a.ry, i);
            V3 M;
            if (a.kernel == RLS_KERNEL_NDF) {
                M = ndf_microfacet(g, rx, ry);
            } else {
                VndfView w = vndf_view(g.view, g.fr, g.ax, g.ay);
                M = vndf_microfacet(w, g.fr, rx, ry);
            }
            st3(a.wi, i, M);
        } else if (OP == OP_NDF_PDF) {
            V3 H = normalize(g.view + ld3(a.cwi, i));
            stg(a.pdf, i, ndf_pdf(g, g.view, H));
        }
    }
}

template <int OP, int FAST_MATH, int MODE>
__global__ RLS_GGX_ATTR(OP) void ggx_kernel(GgxIO a0)
{
    ggx_body<OP, FAST_MATH, MODE>(a0);
}

#if RLS_DIAGNOSTICS
template <int OP, int FAST_MATH, int MODE>
__global__ RLS_GGX_ATTR(OP) void ggx_kernel_stamped(GgxIO a0, unsigned long long *stamps)
{
    ClockStamp<1> cs;
    cs.begin();
    ggx_body<OP, FAST_MATH, MODE>(a0);
    cs.end(stamps);
}
#endif

rls_status check_closure(const rls_ggx_closure *c)
{
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->KsColor), "KsColor planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    return RLS_OK;
}

template <int OP>
rls_status launch_kernel(rls_context *ctx, const GgxIO &io, const char *name)
{
    const rls_ggx_closure &c = io.c;
    const bool streamed = !c.materials.id && c.KsColor.r && c.specularRoughness.v && c.ior.v && c.anisotropic.v;
    const bool uniform = !c.materials.id && !c.specularRoughness.v && !c.ior.v && !c.anisotropic.v;       // specColor: either
#if RLS_DIAGNOSTICS
    if constexpr (OP == OP_REFLECT_REFRACT) {      // BASELINE config 2 under rls_diag_clock_stamps_begin: the stamped instantiation
        if (unsigned long long *stamps = streamed ? rlsh::stamps_for_launch(ctx) : nullptr) {
            hipLaunchKernelGGL((ggx_kernel_stamped<OP, RLS_FAST, STREAMED_ALL>), rlsh::grid_for(ctx, io.n, rlsh::kBlock, RLS_CAP_MULT),
                               dim3(rlsh::kBlock), 0, ctx->stream, io, stamps);
            return rlsh::check_launch(name);
        }
    }
#endif
    if (streamed)
        hipLaunchKernelGGL((ggx_kernel<OP, RLS_FAST, STREAMED_ALL>), rlsh::grid_for(ctx, io.n, rlsh::kBlock, RLS_CAP_MULT), dim3(rlsh::kBlock), 0, ctx->stream, io);
    else if (uniform)   // a thread that hoists wants many tiles to spread the hoisted work over (grid_for_hoisting)
        hipLaunchKernelGGL((ggx_kernel<OP, RLS_FAST, UNIFORM_MATERIAL>), rlsh::grid_for_hoisting(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    else
        hipLaunchKernelGGL((ggx_kernel<OP, RLS_FAST, MIXED>), rlsh::grid_for(ctx, io.n, rlsh::kBlock, RLS_CAP_MULT), dim3(rlsh::kBlock), 0, ctx->stream, io);
    return rlsh::check_launch(name);
}

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_ggx(rls_context *ctx, int op, const rlsh::GgxIO *io)
{
    switch (op) {
    case OP_SAMPLE: return launch_kernel<OP_SAMPLE>(ctx, *io, "rls_ggx_sample[fast]");
    case OP_EVAL: return launch_kernel<OP_EVAL>(ctx, *io, "rls_ggx_eval[fast]");
    case OP_PDF: return launch_kernel<OP_PDF>(ctx, *io, "rls_ggx_pdf[fast]");
    case OP_FUSED: return launch_kernel<OP_FUSED>(ctx, *io, "rls_ggx_sample_eval_pdf[fast]");
    case OP_REFRACT: return launch_kernel<OP_REFRACT>(ctx, *io, "rls_ggx_refract_sample[fast]");
    case OP_REFLECT_REFRACT: return launch_kernel<OP_REFLECT_REFRACT>(ctx, *io, "rls_ggx_reflect_refract[fast]");
    case OP_MICROFACET: return launch_kernel<OP_MICROFACET>(ctx, *io, "rls_ggx_microfacet[fast]");
    default: return launch_kernel<OP_NDF_PDF>(ctx, *io, "rls_ggx_ndf_pdf[fast]");
    }
}
#else
RLS_HIDDEN rls_status rls_fast_ggx(rls_context *ctx, int op, const rlsh::GgxIO *io);

namespace {
template <int OP>
rls_status launch(rls_context *ctx, const GgxIO &io, const char *name)
{
    return ctx->fast ? rls_fast_ggx(ctx, OP, &io) : launch_kernel<OP>(ctx, io, name);
}
} // namespace

#define RLS_PROLOGUE()                                   \
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");          \
    RLS_REQUIRE(n >= 0, "n < 0");                        \
    if (n == 0) return RLS_OK;                           \
    { rls_status _s = check_closure(c); if (_s != RLS_OK) return _s; }

extern "C" {

rls_status rls_ggx_sample(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                          const float *rx, const float *ry, rls_vec3 wi, float *fresnel)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(wi), "wi plane is NULL");
    GgxIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.wi = wi; io.fresnel = fresnel; io.n = n;
    return launch<OP_SAMPLE>(ctx, io, "rls_ggx_sample");
}

rls_status rls_ggx_eval(rls_context *ctx, int64_t n, const rls_ggx_closure *c, rls_cvec3 wi, rls_rgb f)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rlsh::has3(wi) && rlsh::has3(f), "wi/f plane is NULL");
    GgxIO io = {};
    io.c = *c; io.cwi = wi; io.f = f; io.n = n;
    return launch<OP_EVAL>(ctx, io, "rls_ggx_eval");
}

rls_status rls_ggx_pdf(rls_context *ctx, int64_t n, const rls_ggx_closure *c, rls_cvec3 wi, float *pdf)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rlsh::has3(wi) && pdf, "wi/pdf is NULL");
    GgxIO io = {};
    io.c = *c; io.cwi = wi; io.pdf = pdf; io.n = n;
    return launch<OP_PDF>(ctx, io, "rls_ggx_pdf");
}

rls_status rls_ggx_sample_eval_pdf(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                                   const float *rx, const float *ry,
                                   rls_vec3 wi, rls_rgb f, float *pdf, float *fresnel)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(wi) && rlsh::has3(f) && pdf, "wi/f/pdf is NULL");
    GgxIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.wi = wi; io.f = f; io.pdf = pdf; io.fresnel = fresnel; io.n = n;
    return launch<OP_FUSED>(ctx, io, "rls_ggx_sample_eval_pdf");
}

rls_status rls_ggx_refract_sample(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                                  const float *rx, const float *ry,
                                  rls_vec3 wt, float *weight, uint8_t *refracted)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(wt) && weight, "wt/weight is NULL");
    GgxIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.wt = wt; io.weight = weight; io.refracted = refracted; io.n = n;
    return launch<OP_REFRACT>(ctx, io, "rls_ggx_refract_sample");
}

rls_status rls_ggx_reflect_refract(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                                   const float *rx, const float *ry, const float *rx2, const float *ry2,
                                   rls_vec3 wi, rls_rgb f, float *pdf, float *fresnel,
                                   rls_vec3 wt, float *weight)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && ry && rx2 && ry2, "random-number plane is NULL");
    RLS_REQUIRE(rlsh::has3(wi) && rlsh::has3(f) && pdf, "wi/f/pdf is NULL");
    RLS_REQUIRE(rlsh::has3(wt) && weight, "wt/weight is NULL");
    GgxIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.rx2 = rx2; io.ry2 = ry2;
    io.wi = wi; io.f = f; io.pdf = pdf; io.fresnel = fresnel; io.wt = wt; io.weight = weight; io.n = n;
    return launch<OP_REFLECT_REFRACT>(ctx, io, "rls_ggx_reflect_refract");
}

rls_status rls_ggx_microfacet(rls_context *ctx, int64_t n, const rls_ggx_closure *c, int kernel,
                              const float *rx, const float *ry, rls_vec3 m)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(kernel == RLS_KERNEL_VNDF || kernel == RLS_KERNEL_NDF, "unknown sampling kernel");
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(m), "m plane is NULL");
    GgxIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.wi = m; io.kernel = kernel; io.n = n;
    return launch<OP_MICROFACET>(ctx, io, "rls_ggx_microfacet");
}

rls_status rls_ggx_ndf_pdf(rls_context *ctx, int64_t n, const rls_ggx_closure *c, rls_cvec3 wi, float *pdf)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rlsh::has3(wi) && pdf, "wi/pdf is NULL");
    GgxIO io = {};
    io.c = *c; io.cwi = wi; io.pdf = pdf; io.n = n;
    return launch<OP_NDF_PDF>(ctx, io, "rls_ggx_ndf_pdf");
}

} // extern "C"

#endif // !RLS_FAST
