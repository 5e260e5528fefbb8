// scatter.hip -- SssSampler::integrateScatter over an analytic scene (src/rlSss.h:167-280,
// 293-356, 361-424, 439-454).  The probe-ray loop is rls_loops.hpp (scatter_loop); this unit holds the kernel and the
// C-ABI entry point rls_sss_integrate_scatter.  VALU-bound.
#include "rls_loops.hpp"

namespace {

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

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_sss_scatter(rls_context *ctx, int g, const rlsh::ScatterIO *io)
{
    return launch_g(ctx, sss_scatter_kernel<1>, sss_scatter_kernel<4>, sss_scatter_kernel<16>,
                    sss_scatter_kernel<64>, g, *io, "rls_sss_integrate_scatter[fast]");
}
#else
RLS_HIDDEN rls_status rls_fast_sss_scatter(rls_context *ctx, int g, const rlsh::ScatterIO *io);

extern "C" {

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

} // extern "C"

#endif // !RLS_FAST
