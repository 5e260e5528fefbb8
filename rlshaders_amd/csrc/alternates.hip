// alternates.hip -- closures the reference compiles but never selects, kept so that every function
// of its closure files has a batched counterpart: the plain-NDF microfacet samplers of rlDisney
// (sampleGTR2AnisoDirection, sampleGTR2Direction, src/rlDisney.cpp:406-414,504-512), the non-VNDF
// branch of evalSpecularPdf (541-542), D_GTR2 (553-559) and GaussianProfile (src/rlSss.h:63-97).
// Pointwise, HBM-bound, same layout rules as the selected closures.
#include "rls_internal.hpp"

using namespace rlsd;

namespace {

using rlsh::AltIO;
using namespace rlsh;   // AOP_*

__device__ __forceinline__ Disney load_closure(const rls_disney_closure &c, Idx i)
{
    V3 wo = ld3(c.wo, i), N = ld3(c.N, i), T = ld3(c.T, i);
    const PIndex<Idx> k = pindex(c.materials, i);            // parameters by reference (rls_material_index)
    float br, bg, bb;
    ldrgb(c.base_color, k, br, bg, bb);
    float s[10];
    s[0] = ldp(c.subsurface, k); s[1] = ldp(c.metallic, k); s[2] = ldp(c.specular, k);
    s[3] = ldp(c.specular_tint, k); s[4] = ldp(c.roughness, k); s[5] = ldp(c.anisotropic, k);
    s[6] = ldp(c.sheen, k); s[7] = ldp(c.sheen_tint, k); s[8] = ldp(c.clearcoat, k);
    s[9] = ldp(c.clearcoat_gloss, k);
    return disney_make(wo, N, T, br, bg, bb, s);
}

template <int OP, int FAST_MATH = RLS_FAST>
__global__ __launch_bounds__(rlsh::kBlock) void alt_kernel(AltIO a)
{
    stage_libm_tables();
    const TileRange tiles = tile_range(a.n);
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const Idx i = make_idx(base);
        if (i.full() >= a.n) continue;
        if (OP == AOP_LIBM) {
            // one elementary function of the current arithmetic mode, exactly as the closures call it
            const float x = ldg(a.rx, i), y = a.ry ? ldg(a.ry, i) : 0.0f;
            float r, unused;
            switch (a.fn) {
            case RLS_FN_SQRT: r = R_SQRT(x); break;
            case RLS_FN_DIV: r = R_DIV(x, y); break;
#if RLS_FAST   // FAST closures never call atan2f / acosf / tanf (the view analysis is algebraic): ROCm's own libm
            case RLS_FN_ATAN2: r = atan2f(x, y); break;
            case RLS_FN_ACOS: r = acosf(x); break;
            case RLS_FN_TAN: r = tanf(x); break;
            case RLS_FN_SIN: t_sincos(x, &r, &unused); break;
            case RLS_FN_COS: t_sincos(x, &unused, &r); break;
            case RLS_FN_TAN_BOUNDED: r = tanf(x); break;
#else
            case RLS_FN_ATAN2: r = t_atan2(x, y); break;
            case RLS_FN_ACOS: r = t_acos(x); break;
            case RLS_FN_TAN: r = rlm::tan32_q<true>(x); break;
            case RLS_FN_SIN: rlm::sincos32_v<true>(x, &r, &unused); break;
            case RLS_FN_COS: rlm::sincos32_v<true>(x, &unused, &r); break;
            case RLS_FN_TAN_BOUNDED: r = t_tan(x); break;
#endif
            case RLS_FN_SIN_BOUNDED: t_sincos(x, &r, &unused); break;
            case RLS_FN_COS_BOUNDED: t_sincos(x, &unused, &r); break;
            case RLS_FN_EXP: r = R_EXP(x); break;
            case RLS_FN_LOG: r = R_LOG(x); break;
            case RLS_FN_POW: r = y == 5.0f ? R_POW5(x) : R_POW(x, y); break;   // the Schlick exponent takes the closures' own path
            default: r = 0.0f; break;
            }
            stg(a.out1, i, r);
        } else if (OP == AOP_GAUSS) {
            GaussProfile g = gauss_make(ldp(a.dist_x, i));
            float r = gauss_radius(g, ldg(a.rx, i));
            stg(a.r, i, r);
            stg(a.pdf, i, gauss_pdf(g, r));
            stg(a.profile, i, gauss_profile(g, r));
        } else {
            Disney d = load_closure(a.c, i);
            if (OP == AOP_GTR2_ANISO) st3(a.out3, i, disney_gtr2_aniso_microfacet(d, ldg(a.rx, i), ldg(a.ry, i)));
            else if (OP == AOP_GTR2) st3(a.out3, i, disney_gtr2_direction(d, ldg(a.rx, i), ldg(a.ry, i)));
            else if (OP == AOP_NDF_PDF) stg(a.out1, i, disney_specular_pdf_ndf(d, ld3(a.v, i)));
            else stg(a.out1, i, D_GTR2(d, ld3(a.v, i)));
        }
    }
}

template <int OP>
rls_status launch_kernel(rls_context *ctx, const AltIO &io, const char *name)
{
    hipLaunchKernelGGL(alt_kernel<OP>, rlsh::grid_for(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    return rlsh::check_launch(name);
}

rls_status dispatch(rls_context *ctx, int op, const AltIO &io, const char *name)
{
    switch (op) {
    case AOP_GTR2_ANISO: return launch_kernel<AOP_GTR2_ANISO>(ctx, io, name);
    case AOP_GTR2: return launch_kernel<AOP_GTR2>(ctx, io, name);
    case AOP_NDF_PDF: return launch_kernel<AOP_NDF_PDF>(ctx, io, name);
    case AOP_D_GTR2: return launch_kernel<AOP_D_GTR2>(ctx, io, name);
    case AOP_LIBM: return launch_kernel<AOP_LIBM>(ctx, io, name);
    default: return launch_kernel<AOP_GAUSS>(ctx, io, name);
    }
}

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_alt(rls_context *ctx, int op, const rlsh::AltIO *io) { return dispatch(ctx, op, *io, "alternates[fast]"); }
#else
RLS_HIDDEN rls_status rls_fast_alt(rls_context *ctx, int op, const rlsh::AltIO *io);

namespace {
rls_status run(rls_context *ctx, int op, const AltIO &io, const char *name)
{
    return ctx->fast ? rls_fast_alt(ctx, op, &io) : dispatch(ctx, op, io, name);
}
rls_status check_closure(const rls_disney_closure *c)
{
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->base_color), "base_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    return RLS_OK;
}
} // namespace

#define RLS_PROLOGUE()                                   \
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");          \
    RLS_REQUIRE(n >= 0, "n < 0");                        \
    if (n == 0) return RLS_OK;

extern "C" {

rls_status rls_disney_alt_sample(rls_context *ctx, int64_t n, const rls_disney_closure *c, int kind,
                                 const float *rx, const float *ry, rls_vec3 m)
{
    RLS_PROLOGUE();
    { rls_status s = check_closure(c); if (s != RLS_OK) return s; }
    RLS_REQUIRE(kind == RLS_DISNEY_ALT_GTR2_ANISO || kind == RLS_DISNEY_ALT_GTR2, "unknown alternate sampler");
    RLS_REQUIRE(rx && ry && rlsh::has3(m), "NULL plane");
    AltIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.out3 = m; io.n = n;
    return run(ctx, kind == RLS_DISNEY_ALT_GTR2_ANISO ? AOP_GTR2_ANISO : AOP_GTR2, io, "rls_disney_alt_sample");
}

rls_status rls_disney_alt_pdf(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 wi, float *pdf)
{
    RLS_PROLOGUE();
    { rls_status s = check_closure(c); if (s != RLS_OK) return s; }
    RLS_REQUIRE(rlsh::has3(wi) && pdf, "NULL plane");
    AltIO io = {};
    io.c = *c; io.v = wi; io.out1 = pdf; io.n = n;
    return run(ctx, AOP_NDF_PDF, io, "rls_disney_alt_pdf");
}

rls_status rls_disney_d_gtr2(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 m, float *d)
{
    RLS_PROLOGUE();
    { rls_status s = check_closure(c); if (s != RLS_OK) return s; }
    RLS_REQUIRE(rlsh::has3(m) && d, "NULL plane");
    AltIO io = {};
    io.c = *c; io.v = m; io.out1 = d; io.n = n;
    return run(ctx, AOP_D_GTR2, io, "rls_disney_d_gtr2");
}

rls_status rls_gaussian_sample(rls_context *ctx, int64_t n, rls_param dist_x, const float *rx,
                               float *r, float *pdf, float *profile)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(rx && r && pdf && profile, "NULL plane");
    AltIO io = {};
    io.dist_x = dist_x; io.rx = rx; io.r = r; io.pdf = pdf; io.profile = profile; io.n = n;
    return run(ctx, AOP_GAUSS, io, "rls_gaussian_sample");
}

rls_status rls_libm_eval(rls_context *ctx, int fn, int64_t n, const float *x, const float *y, float *out)
{
    RLS_PROLOGUE();
    RLS_REQUIRE(fn >= RLS_FN_SQRT && fn <= RLS_FN_COS_BOUNDED, "unknown function id");
    const bool binary = fn == RLS_FN_DIV || fn == RLS_FN_ATAN2 || fn == RLS_FN_POW;
    RLS_REQUIRE(x && out && (!binary || y), "NULL plane");
    AltIO io = {};
    io.rx = x; io.ry = binary ? y : nullptr; io.out1 = out; io.fn = fn; io.n = n;
    return run(ctx, AOP_LIBM, io, "rls_libm_eval");
}

} // extern "C"

#endif // !RLS_FAST
