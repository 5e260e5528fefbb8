// lights.hip -- the light loops of rlGgx and rlDisney (src/rlGgx.cpp:274-299,
// src/rlDisney.cpp:695-705): two-sample MIS over up to eight spherical lights per shading point.  The loops are
// rls_loops.hpp (ggx_direct_loops, disney_direct_loops); this unit holds the kernels and the C-ABI entry points
// rls_ggx_direct_lighting / rls_disney_direct_lighting.  VALU-bound.
#include "rls_loops.hpp"

namespace {

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

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_ggx_direct(rls_context *ctx, int g, const rlsh::LightIO *io)
{
    return launch_g(ctx, ggx_direct_kernel<1>, ggx_direct_kernel<4>, ggx_direct_kernel<16>,
                    ggx_direct_kernel<64>, g, *io, "rls_ggx_direct_lighting[fast]");
}
RLS_HIDDEN rls_status rls_fast_disney_direct(rls_context *ctx, int g, const rlsh::DisneyLightIO *io)
{
    return launch_g(ctx, disney_direct_kernel<1>, disney_direct_kernel<4>, disney_direct_kernel<16>,
                    disney_direct_kernel<64>, g, *io, "rls_disney_direct_lighting[fast]");
}
#else
RLS_HIDDEN rls_status rls_fast_ggx_direct(rls_context *ctx, int g, const rlsh::LightIO *io);
RLS_HIDDEN rls_status rls_fast_disney_direct(rls_context *ctx, int g, const rlsh::DisneyLightIO *io);

extern "C" {

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
