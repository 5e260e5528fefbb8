// disney.hip -- rlDisney closure kernels (DisneySampler, src/rlDisney.cpp:105-602 of the
// reference) and their C-ABI entry points.  gfx950, wave64, one shading point per lane.
//
// Roofline: HBM for the 1-sample calls.  Algorithmic bytes per point with every parameter
// streamed: 88 B of closure (wo3 N3 T3 base3 + 10 scalars) + 8 B xi in, 28 B out (wi3 f3 pdf).
#include "rls_internal.hpp"

using namespace rlsd;

namespace {

using rlsh::DisneyIO;

#define RLS_DISNEY_ARGS(a0) reload_args(a0)
enum { OP_SAMPLE = rlsh::DOP_SAMPLE, OP_EVAL = rlsh::DOP_EVAL, OP_PDF = rlsh::DOP_PDF, OP_FUSED = rlsh::DOP_FUSED };

// MODE (checked on the host): STREAMED_ALL every parameter is a per-point plane; UNIFORM_ALL every one is a single value for
// the batch (an Arnold parameter is a constant unless a texture is linked to it) and the parameter-only arithmetic -- the
// constructor (tint, F0, sheen colour, aspect / alpha_x / alpha_y: src/rlDisney.cpp:155-192) and the clearcoat terms the
// verbs recompute per call (logf(a2), clearcoat / (clearcoat + 1)) -- runs once per thread ahead of the tile loop, its results
// kept in scalar registers; UNIFORM_SCALARS the same with base_color as per-point planes (a colour map on an otherwise plain
// node): the half of the constructor that reads base_color stays per point; MIXED tests parameter by parameter in the loop
enum { MIXED = 0, STREAMED_ALL = 1, UNIFORM_ALL = 2, UNIFORM_SCALARS = 3 };

template <bool WITH_BASE>
__device__ __forceinline__ Disney uniform_closure(const rls_disney_closure &c, DisneyTints &t)
{
    const float s[10] = { c.subsurface.u, c.metallic.u, c.specular.u, c.specular_tint.u, c.roughness.u, c.anisotropic.u,
                          c.sheen.u, c.sheen_tint.u, c.clearcoat.u, c.clearcoat_gloss.u };
    Disney d = {};
    t = disney_make_scalars(d, s);
    if (WITH_BASE) disney_make_base(d, t, c.base_color.ur, c.base_color.ug, c.base_color.ub);
    disney_prepare_material(d);
    disney_wave_uniform(d);
    t = disney_wave_uniform(t);
    return d;
}

template <bool STREAMED>
__device__ __forceinline__ Disney load_closure(const rls_disney_closure &c, Idx i)
{
    V3 wo = ld3(c.wo, i), N = ld3(c.N, i), T = ld3(c.T, i);
    const PIndex<Idx> k = pindex<!STREAMED>(c.materials, i);           // parameters by reference run the MIXED kernel
    float br, bg, bb;
    ldrgb<STREAMED>(c.base_color, k, br, bg, bb);
    float s[10];
    s[0] = ldp<STREAMED>(c.subsurface, k);
    s[1] = ldp<STREAMED>(c.metallic, k);
    s[2] = ldp<STREAMED>(c.specular, k);
    s[3] = ldp<STREAMED>(c.specular_tint, k);
    s[4] = ldp<STREAMED>(c.roughness, k);
    s[5] = ldp<STREAMED>(c.anisotropic, k);
    s[6] = ldp<STREAMED>(c.sheen, k);
    s[7] = ldp<STREAMED>(c.sheen_tint, k);
    s[8] = ldp<STREAMED>(c.clearcoat, k);
    s[9] = ldp<STREAMED>(c.clearcoat_gloss, k);
    return disney_make(wo, N, T, br, bg, bb, s);
}

// occupancy of the rlDisney kernels (waves per SIMD the register allocator must allow).  Left alone the glossy triple takes 86
// vector registers (five waves); pinned at six (80 registers, nothing spilled) it runs 1.7 % faster, the colour-map form 2.1 %
// (2.080 -> 2.045 ms, 1.794 -> 1.756 ms; tools/ab.sh, two interleaved repetitions); at seven it spills (-0.6 %), at four +3 %
#ifndef RLS_DISNEY_WAVES
#define RLS_DISNEY_WAVES 6
#endif
#define RLS_DISNEY_ATTR __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_DISNEY_WAVES, RLS_DISNEY_WAVES)))
template <int OP, bool DIFFUSE, int FAST_MATH, int MODE>
__global__ RLS_DISNEY_ATTR void disney_kernel(DisneyIO a0)
{
    stage_libm_tables();   // powf / logf tables -> LDS (EXACT mode)
    Disney ud = {};
    DisneyTints ut = {};
    if (MODE == UNIFORM_ALL) ud = uniform_closure<true>(a0.c, ut);
    if (MODE == UNIFORM_SCALARS) ud = uniform_closure<false>(a0.c, ut);
    const TileRange tiles = tile_range(a0.n);
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const Idx i = make_idx(base);
        if (i.full() >= a0.n) continue;
        // plane pointers re-read from the kernarg segment where they are used (rls_internal.hpp, reload_args)
        const DisneyIO a = RLS_DISNEY_ARGS(a0);
        Disney d;
        if (MODE == UNIFORM_ALL || MODE == UNIFORM_SCALARS) {
            d = ud;
            d.view = ld3(a.c.wo, i);
            d.fr.N = ld3(a.c.N, i);
            d.fr.U = ld3(a.c.T, i);
            d.fr.V = cross(d.fr.N, d.fr.U);
            if (MODE == UNIFORM_SCALARS) {
                float br, bg, bb;
                ldrgb<true>(a.c.base_color, i, br, bg, bb);
                disney_make_base(d, ut, br, bg, bb);
            }
            disney_prepare_view(d);
        } else {
            d = load_closure<MODE == STREAMED_ALL>(a.c, i);
            disney_prepare(d);
        }
        V3 L;
        if (OP == OP_SAMPLE || OP == OP_FUSED) {
            float rx = ldg(a.rx, i), ry = ldg(a.ry, i);
            if (DIFFUSE) {
                L = cosine_hemisphere(d.fr, rx, ry);                  // src/rlDisney.cpp:359-365
            } else {
                VndfView w = vndf_view(d.view, d.fr, d.ax, d.ay);
                L = disney_sample_specular(d, w, rx, ry);             // src/rlDisney.cpp:367-390
            }
            const DisneyIO b = RLS_DISNEY_ARGS(a0);
            st3(b.wi, i, L);
        } else {
            L = ld3(a.cwi, i);
        }
        if (OP == OP_EVAL || OP == OP_PDF || OP == OP_FUSED) {
            float r, g, b, pdf;
            disney_eval_pdf<DIFFUSE, OP != OP_PDF, OP != OP_EVAL>(d, L, r, g, b, pdf);
            const DisneyIO o = RLS_DISNEY_ARGS(a0);
            if (OP != OP_PDF) strgb(o.f, i, r, g, b);
            if (OP != OP_EVAL) stg(o.pdf, i, pdf);
        }
    }
}

rls_status check_closure(const rls_disney_closure *c, int lobe)
{
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(lobe == RLS_RAY_DIFFUSE || lobe == RLS_RAY_GLOSSY, "lobe must be RLS_RAY_DIFFUSE or RLS_RAY_GLOSSY");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->base_color), "base_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    return RLS_OK;
}

template <int OP>
rls_status launch_kernel(rls_context *ctx, int lobe, const DisneyIO &io, const char *name)
{
    const rls_disney_closure &c = io.c;
    const bool by_reference = c.materials.id != nullptr;
    const bool streamed = !by_reference && c.base_color.r && c.subsurface.v && c.metallic.v && c.specular.v && c.specular_tint.v &&
                          c.roughness.v && c.anisotropic.v && c.sheen.v && c.sheen_tint.v && c.clearcoat.v &&
                          c.clearcoat_gloss.v;
    const bool scalars = !by_reference && !c.subsurface.v && !c.metallic.v && !c.specular.v && !c.specular_tint.v && !c.roughness.v &&
                         !c.anisotropic.v && !c.sheen.v && !c.sheen_tint.v && !c.clearcoat.v && !c.clearcoat_gloss.v;
    const bool uniform = scalars && !c.base_color.r, colour_map = scalars && c.base_color.r;
    const dim3 grid = scalars ? rlsh::grid_for_hoisting(ctx, io.n) : rlsh::grid_for(ctx, io.n);
    const dim3 block(rlsh::kBlock);
    if (lobe == RLS_RAY_DIFFUSE) {
        if (streamed) hipLaunchKernelGGL((disney_kernel<OP, true, RLS_FAST, STREAMED_ALL>), grid, block, 0, ctx->stream, io);
        else if (uniform) hipLaunchKernelGGL((disney_kernel<OP, true, RLS_FAST, UNIFORM_ALL>), grid, block, 0, ctx->stream, io);
        else if (colour_map) hipLaunchKernelGGL((disney_kernel<OP, true, RLS_FAST, UNIFORM_SCALARS>), grid, block, 0, ctx->stream, io);
        else hipLaunchKernelGGL((disney_kernel<OP, true, RLS_FAST, MIXED>), grid, block, 0, ctx->stream, io);
    } else {
        if (streamed) hipLaunchKernelGGL((disney_kernel<OP, false, RLS_FAST, STREAMED_ALL>), grid, block, 0, ctx->stream, io);
        else if (uniform) hipLaunchKernelGGL((disney_kernel<OP, false, RLS_FAST, UNIFORM_ALL>), grid, block, 0, ctx->stream, io);
        else if (colour_map) hipLaunchKernelGGL((disney_kernel<OP, false, RLS_FAST, UNIFORM_SCALARS>), grid, block, 0, ctx->stream, io);
        else hipLaunchKernelGGL((disney_kernel<OP, false, RLS_FAST, MIXED>), grid, block, 0, ctx->stream, io);
    }
    return rlsh::check_launch(name);
}

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_disney(rls_context *ctx, int op, int lobe, const rlsh::DisneyIO *io)
{
    switch (op) {
    case OP_SAMPLE: return launch_kernel<OP_SAMPLE>(ctx, lobe, *io, "rls_disney_sample[fast]");
    case OP_EVAL: return launch_kernel<OP_EVAL>(ctx, lobe, *io, "rls_disney_eval[fast]");
    case OP_PDF: return launch_kernel<OP_PDF>(ctx, lobe, *io, "rls_disney_pdf[fast]");
    default: return launch_kernel<OP_FUSED>(ctx, lobe, *io, "rls_disney_sample_eval_pdf[fast]");
    }
}
#else
RLS_HIDDEN rls_status rls_fast_disney(rls_context *ctx, int op, int lobe, const rlsh::DisneyIO *io);

namespace {
template <int OP>
rls_status launch(rls_context *ctx, int lobe, const DisneyIO &io, const char *name)
{
    return ctx->fast ? rls_fast_disney(ctx, OP, lobe, &io) : launch_kernel<OP>(ctx, lobe, io, name);
}
} // namespace

#define RLS_PROLOGUE()                                   \
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");          \
    RLS_REQUIRE(n >= 0, "n < 0");                        \
    if (n == 0) return RLS_OK;                           \
    { rls_status _s = check_closure(c, lobe); if (_s != RLS_OK) return _s; }

extern "C" {

rls_status rls_disney_sample(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                             const float *rx, const float *ry, rls_vec3 wi)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(wi), "wi plane is NULL");
    DisneyIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.wi = wi; io.n = n;
    return launch<OP_SAMPLE>(ctx, lobe, io, "rls_disney_sample");
}

rls_status rls_disney_eval(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                           rls_cvec3 wi, rls_rgb f)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rlsh::has3(wi) && rlsh::has3(f), "wi/f plane is NULL");
    DisneyIO io = {};
    io.c = *c; io.cwi = wi; io.f = f; io.n = n;
    return launch<OP_EVAL>(ctx, lobe, io, "rls_disney_eval");
}

rls_status rls_disney_pdf(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                          rls_cvec3 wi, float *pdf)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rlsh::has3(wi) && pdf, "wi/pdf is NULL");
    DisneyIO io = {};
    io.c = *c; io.cwi = wi; io.pdf = pdf; io.n = n;
    return launch<OP_PDF>(ctx, lobe, io, "rls_disney_pdf");
}

rls_status rls_disney_sample_eval_pdf(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                                      const float *rx, const float *ry,
                                      rls_vec3 wi, rls_rgb f, float *pdf)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(wi) && rlsh::has3(f) && pdf, "wi/f/pdf is NULL");
    DisneyIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.wi = wi; io.f = f; io.pdf = pdf; io.n = n;
    return launch<OP_FUSED>(ctx, lobe, io, "rls_disney_sample_eval_pdf");
}

} // extern "C"

#endif // !RLS_FAST
