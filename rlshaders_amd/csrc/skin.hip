// skin.hip -- rlSkin composite kernel: sheen GGX lobe + specular GGX lobe + NDProfile SSS sample
// with the layer-weight arithmetic of shader_evaluate (src/rlSkin.cpp:174-246 of the reference).
// One (sample, eval, pdf) triple per GGX lobe stands in for Arnold's light loop/AiBRDFIntegrate;
// the Fresnel average the reference hands down between layers (src/rlSkin.cpp:204,228) is the
// Fresnel term of that sample.  gfx950, one shading point per lane.
//
// Roofline: HBM.  Algorithmic bytes per point (every parameter streamed): 140 B in (wo3 N3 T3,
// sss_color3 dist3 mult w_sss, spec col3 w rough ior, sheen col3 w rough ior, xi6) + 96 B out
// (2 x [wi3 f3 pdf F] + [r pdf R3] + [sheenFresnel specularFresnel sssWeight]) = 236 B = 3 samples
// (SURVEY.md section 8(d), config 5).
// loads behind reload_args / the per-parameter stream-or-uniform branches sit in later basic blocks than make_idx():
// they renew the lane-offset barrier (rls_device.hpp) so that every plane access keeps the scalar-base addressing form
#ifndef RLS_LOAD_RENEW
#define RLS_LOAD_RENEW 1
#endif
#include "rls_internal.hpp"

using namespace rlsd;

namespace {

using rlsh::SkinIO;

struct LobeOut { V3 wi; float fr, fg, fb, pdf, F; };

// the part of a lobe after the microfacet normal: reflect, Fresnel side effect, evalBrdf, evalPdf
__device__ __forceinline__ LobeOut lobe_from_microfacet(const Ggx &g, V3 M)
{
    LobeOut o;
    o.wi = reflect_direction(g.view, M);
    o.F = ggx_fresnel(g, o.wi, M);
    ggx_eval_pdf<true, true>(g, o.wi, o.fr, o.fg, o.fb, o.pdf);
    return o;
}

// One isotropic GGX lobe of rlSkin (src/rlSkin.cpp:192,215: anisotropic defaulted to 0).
__device__ __forceinline__ LobeOut ggx_lobe(V3 wo, V3 N, V3 T, V3 local, const GgxMaterial &m, float cr, float cg, float cb,
                                            float rx, float ry)
{
    LobeOut o;
    Ggx g = ggx_from_material(m, wo, N, T, false, cr, cg, cb);
    // `local` = the view in the shared (T, N x T, N) frame: the same for both lobes, computed once
    VndfView w = vndf_view_from(local, g.ax, g.ay);
    V3 M = vndf_microfacet(w, g.fr, rx, ry);
    o.wi = reflect_direction(g.view, M);
    o.F = ggx_fresnel(g, o.wi, M);
    ggx_eval_pdf<true, true>(g, o.wi, o.fr, o.fg, o.fb, o.pdf);
    return o;
}

// MODE (checked on the host): STREAMED every parameter is a per-point plane; UNIFORM every parameter is one value for the batch
// (an Arnold parameter is a constant unless a texture is linked to it): the parameter-only arithmetic -- the two lobes'
// roughness / ior terms, NDProfile::setDistance with its six expf -- runs once per thread ahead of the tile loop and stays in
// scalar registers -- with the layer weights uniform as well the gates of the three layers are scalar branches; MIXED tests
// parameter by parameter in the loop.  (Measured and not kept: a relaxed form that lets the colours and weights be planes beside
// hoisted roughness / ior / distance terms -- the colour-map case -- gains 1 % over STREAMED there and costs the all-uniform case
// 6 %: 3.47 -> 3.73 ms.)
enum { MIXED = 0, STREAMED_ALL = 1, UNIFORM_ALL = 2 };
// (hoisted values -- the lobes' and NDProfile's -- all move to scalar registers, and the hoisted profile keeps all of getPdf's
// reciprocals; the per-point profile of the streamed kernel keeps the three of d_i only, as in sss.hip)
//
// Occupancy of the rlSkin kernel (waves per SIMD the register allocator must allow).  Left alone it takes 95 vector registers --
// five waves, one register pair short of four: NDProfile's single range tests (rls_device.hpp, nd_make / nd_pdf_profile_t),
// which gain 2-7 % in the rlSss kernels, pushed it to 98-101 registers and four waves, +4.5 ... +5.6 %.
// Pinned at six waves (80 registers, 4 of them spilled: 20 B of scratch per lane) with those tests: 4.086 -> 3.997 ms (-2.2 %),
// uniform parameters -0.8 %; at five -1.1 %, at seven +3.7 % (profiles/r03_exp_range_once.txt).
#ifndef RLS_SKIN_WAVES
#define RLS_SKIN_WAVES 6
#endif
#define RLS_SKIN_ATTR __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_SKIN_WAVES, RLS_SKIN_WAVES)))
// the kernel body: inlined into skin_kernel (the product) and skin_kernel_stamped (diagnostic: the same body between clock
// stamps, rls_internal.hpp ClockStamp).  a0 is the kernel's first parameter (reload_args).
template <int FAST_MATH, int MODE>
__device__ __forceinline__ void skin_body(const SkinIO &a0)
{
    constexpr bool STREAMED = MODE == STREAMED_ALL, UNIFORM = MODE == UNIFORM_ALL;
    stage_libm_tables();   // expf / logf tables -> LDS (EXACT mode)
    GgxMaterial um1 = {}, um2 = {};
    NdProfile up = {};
    if (UNIFORM) {
        const rls_skin_closure &c = a0.c;
        um1 = ggx_material<true>(c.sheen_ior.u, c.sheen_roughness.u, 0.0f);
        um2 = ggx_material<true>(c.specular_ior.u, c.specular_roughness.u, 0.0f);
        const float mult = c.sss_dist_multiplier.u;                               // src/rlSkin.cpp:235-236
        up = nd_make<true>(c.sss_scatter_dist[0].u * mult, c.sss_scatter_dist[1].u * mult, c.sss_scatter_dist[2].u * mult);
        um1 = ggx_material_wave_uniform(um1); um2 = ggx_material_wave_uniform(um2);
        up = nd_wave_uniform(up);
    }
    const TileRange tiles = tile_range(a0.n);
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const Idx i = make_idx(base);
        if (i.full() >= a0.n) continue;
        const SkinIO a = reload_args(a0);      // the 35 input planes' pointers, for the loads of this tile only
        const rls_skin_closure &c = a.c;
        const PIndex<Idx> pk = pindex<MODE == MIXED>(c.materials, i);         // parameters by reference run the MIXED kernel
#define LDP(param) (UNIFORM ? (param).u : ldp<STREAMED>(param, pk))
        V3 wo = ld3(c.wo, i), N = ld3(c.N, i), T = ld3(c.T, i);

        float sheenFresnel = 0.0f, specularFresnel = 0.0f;
        LobeOut sh = {}, sp = {};
        Frame gfr;
        gfr.N = N; gfr.U = T; gfr.V = cross(N, T);
        V3 local = vndf_local(wo, gfr);

        float sheenWeight = LDP(c.sheen_weight);
        float shr = 0.0f, shg = 0.0f, shb = 0.0f;
        if (UNIFORM) { shr = c.sheen_color.ur; shg = c.sheen_color.ug; shb = c.sheen_color.ub; }
        else ldrgb<STREAMED>(c.sheen_color, pk, shr, shg, shb);
        float sheenIor = LDP(c.sheen_ior), sheenRough = LDP(c.sheen_roughness);
        float rx0 = ldg(a.xi[0], i), ry0 = ldg(a.xi[1], i);
        float specWeight = LDP(c.specular_weight);
        float spr = 0.0f, spg = 0.0f, spb = 0.0f;
        if (UNIFORM) { spr = c.specular_color.ur; spg = c.specular_color.ug; spb = c.specular_color.ub; }
        else ldrgb<STREAMED>(c.specular_color, pk, spr, spg, spb);
        float specIor = LDP(c.specular_ior), specRough = LDP(c.specular_roughness);
        float rx1 = ldg(a.xi[2], i), ry1 = ldg(a.xi[3], i);
        const bool sheenOn = sheenWeight > kEps, specOn = specWeight > kEps;          // src/rlSkin.cpp:191, 214
        const GgxMaterial m1 = UNIFORM ? um1 : ggx_material<true>(sheenIor, sheenRough, 0.0f);
        const GgxMaterial m2 = UNIFORM ? um2 : ggx_material<true>(specIor, specRough, 0.0f);
        if (UNIFORM ? (sheenOn && specOn) : __builtin_amdgcn_ballot_w64(sheenOn && specOn) == ~0ull) {
            // every lane of the wavefront evaluates both lobes: their two microfacet samples share one pass of the
            // uniform-slope fallback (vndf_microfacet_pair)
            Ggx g1 = ggx_from_material(m1, wo, N, T, false, shr, shg, shb);
            Ggx g2 = ggx_from_material(m2, wo, N, T, false, spr, spg, spb);
            VndfView w1 = vndf_view_from(local, g1.ax, g1.ay), w2 = vndf_view_from(local, g2.ax, g2.ay);
            V3 M1, M2;
            vndf_microfacet_pair(w1, g1.fr, rx0, ry0, w2, g2.fr, rx1, ry1, M1, M2);
            sh = lobe_from_microfacet(g1, M1);
            sp = lobe_from_microfacet(g2, M2);
            sheenFresnel = sh.F * sheenWeight;                              // :204 (one sample)
            specularFresnel = sp.F * specWeight;                            // :228
        } else
        {
            if (sheenOn) {
                sh = ggx_lobe(wo, N, T, local, m1, shr, shg, shb, rx0, ry0);
                sheenFresnel = sh.F * sheenWeight;
            }
            if (specOn) {
                sp = ggx_lobe(wo, N, T, local, m2, spr, spg, spb, rx1, ry1);
                specularFresnel = sp.F * specWeight;
            }
        }

        float mult = LDP(c.sss_dist_multiplier);                         // :235-236
        float dx = LDP(c.sss_scatter_dist[0]) * mult;
        float dy = LDP(c.sss_scatter_dist[1]) * mult;
        float dz = LDP(c.sss_scatter_dist[2]) * mult;
        float sssWeight = LDP(c.sss_weight);
        sssWeight *= 1.0f - specularFresnel * (1.0f - sheenFresnel);        // :238
        float rx2 = ldg(a.xi[4], i), ry2 = ldg(a.xi[5], i);

        float r = 0.0f, rpdf = 0.0f, R = 0.0f, G = 0.0f, B = 0.0f;
        if (!(sssWeight < kEps)) {                                          // :244
            const NdProfile p = UNIFORM ? up : nd_make<false>(dx, dy, dz);
            Frame fr = sss_frame(N, T, true);                               // src/rlSss.h:151-154
            V3 off, dir;
            float maxdist;
            r = sss_probe_ray(p, fr, rx2, ry2, off, dir, maxdist);
            nd_pdf_profile(p, r, rpdf, R, G, B);
        }

        const SkinIO b = reload_args(a0);      // ... and the 24 output planes', for its stores
        const rls_skin_out &o = b.o;
        st3(o.sheen_wi, i, sh.wi); strgb(o.sheen_f, i, sh.fr, sh.fg, sh.fb);
        stg(o.sheen_pdf, i, sh.pdf); stg(o.sheen_fresnel, i, sh.F);
        st3(o.spec_wi, i, sp.wi); strgb(o.spec_f, i, sp.fr, sp.fg, sp.fb);
        stg(o.spec_pdf, i, sp.pdf); stg(o.spec_fresnel, i, sp.F);
        stg(o.r, i, r); stg(o.r_pdf, i, rpdf); strgb(o.profile, i, R, G, B);
        stg(o.sheenFresnel, i, sheenFresnel);
        stg(o.specularFresnel, i, specularFresnel);
        stg(o.sssWeight, i, sssWeight);
#undef LDP
    }
}

template <int FAST_MATH, int MODE>
__global__ RLS_SKIN_ATTR void skin_kernel(SkinIO a0)
{
    skin_body<FAST_MATH, MODE>(a0);
}

#if RLS_DIAGNOSTICS
template <int FAST_MATH, int MODE>
__global__ RLS_SKIN_ATTR void skin_kernel_stamped(SkinIO a0, unsigned long long *stamps)
{
    ClockStamp<1> cs;
    cs.begin();
    skin_body<FAST_MATH, MODE>(a0);
    cs.end(stamps);
}
#endif

rls_status launch_kernel(rls_context *ctx, const SkinIO &io, const char *name)
{
    const rls_skin_closure &c = io.c;
    const bool by_reference = c.materials.id != nullptr;
    const bool streamed = !by_reference && c.sss_color.r && c.sss_weight.v && c.sss_dist_multiplier.v && c.sss_scatter_dist[0].v &&
                          c.sss_scatter_dist[1].v && c.sss_scatter_dist[2].v && c.specular_color.r &&
                          c.specular_weight.v && c.specular_roughness.v && c.specular_ior.v && c.sheen_color.r &&
                          c.sheen_weight.v && c.sheen_roughness.v && c.sheen_ior.v;
    const bool uniform = !by_reference && !c.sss_color.r && !c.sss_weight.v && !c.sss_dist_multiplier.v && !c.sss_scatter_dist[0].v &&
                         !c.sss_scatter_dist[1].v && !c.sss_scatter_dist[2].v && !c.specular_color.r &&
                         !c.specular_weight.v && !c.specular_roughness.v && !c.specular_ior.v && !c.sheen_color.r &&
                         !c.sheen_weight.v && !c.sheen_roughness.v && !c.sheen_ior.v;
    const dim3 grid = rlsh::grid_for(ctx, io.n, rlsh::kBlock, RLS_CAP_MULT);
#if RLS_DIAGNOSTICS
    if (unsigned long long *stamps = streamed ? rlsh::stamps_for_launch(ctx) : nullptr) {
        // BASELINE config 5 under rls_diag_clock_stamps_begin: the stamped instantiation
        hipLaunchKernelGGL((skin_kernel_stamped<RLS_FAST, STREAMED_ALL>), grid, dim3(rlsh::kBlock), 0, ctx->stream, io, stamps);
        return rlsh::check_launch(name);
    }
#endif
    if (streamed)
        hipLaunchKernelGGL((skin_kernel<RLS_FAST, STREAMED_ALL>), grid, dim3(rlsh::kBlock), 0, ctx->stream, io);
    else if (uniform)   // a thread that hoists wants many tiles to spread the hoisted work over (grid_for_hoisting)
        hipLaunchKernelGGL((skin_kernel<RLS_FAST, UNIFORM_ALL>), rlsh::grid_for_hoisting(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    else
        hipLaunchKernelGGL((skin_kernel<RLS_FAST, MIXED>), grid, dim3(rlsh::kBlock), 0, ctx->stream, io);
    return rlsh::check_launch(name);
}

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_skin(rls_context *ctx, const rlsh::SkinIO *io)
{
    return launch_kernel(ctx, *io, "rls_skin_sample_eval_pdf[fast]");
}
#else
RLS_HIDDEN rls_status rls_fast_skin(rls_context *ctx, const rlsh::SkinIO *io);

extern "C" {

rls_status rls_skin_sample_eval_pdf(rls_context *ctx, int64_t n, const rls_skin_closure *c,
                                    const float *const xi[6], const rls_skin_out *out)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(c != nullptr && xi != nullptr && out != nullptr, "NULL argument");
    RLS_REQUIRE(rlsh::has3(c->wo) && rlsh::has3(c->N) && rlsh::has3(c->T), "wo/N/T plane is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->sss_color) && rlsh::ok_rgb(c->specular_color) && rlsh::ok_rgb(c->sheen_color),
                "colour planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    for (int k = 0; k < 6; k++) RLS_REQUIRE(xi[k] != nullptr, "xi plane is NULL");
    RLS_REQUIRE(rlsh::has3(out->sheen_wi) && rlsh::has3(out->sheen_f) && out->sheen_pdf && out->sheen_fresnel &&
                rlsh::has3(out->spec_wi) && rlsh::has3(out->spec_f) && out->spec_pdf && out->spec_fresnel &&
                out->r && out->r_pdf && rlsh::has3(out->profile) &&
                out->sheenFresnel && out->specularFresnel && out->sssWeight, "NULL output plane");
    SkinIO io = {};
    io.c = *c;
    for (int k = 0; k < 6; k++) io.xi[k] = xi[k];
    io.o = *out;
    io.n = n;
    return ctx->fast ? rls_fast_skin(ctx, &io) : launch_kernel(ctx, io, "rls_skin_sample_eval_pdf");
}

} // extern "C"

#endif // !RLS_FAST
