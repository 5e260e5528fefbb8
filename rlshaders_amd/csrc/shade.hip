// shade.hip -- the three nodes' whole shader_evaluate in one launch (beyond SURVEY.md section 8: src/rlGgx.cpp:248-327,
// src/rlDisney.cpp:685-727, src/rlSkin.cpp:174-254): the loops of rls_loops.hpp run back to back on one closure set-up.
// Kernels and C-ABI entry points rls_ggx_shade / rls_disney_shade / rls_skin_integrate.  VALU-bound; tested, not
// developed further.
#include "rls_loops.hpp"

namespace {

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

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_skin_integrate(rls_context *ctx, int g, const rlsh::SkinIntIO *io)
{
    return launch_g(ctx, skin_integrate_kernel<1>, skin_integrate_kernel<4>, skin_integrate_kernel<16>,
                    skin_integrate_kernel<64>, g, *io, "rls_skin_integrate[fast]");
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
#else
RLS_HIDDEN rls_status rls_fast_skin_integrate(rls_context *ctx, int g, const rlsh::SkinIntIO *io);
RLS_HIDDEN rls_status rls_fast_ggx_shade(rls_context *ctx, int g, const rlsh::GgxShadeIO *io);
RLS_HIDDEN rls_status rls_fast_disney_shade(rls_context *ctx, int g, const rlsh::DisneyShadeIO *io);

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

} // extern "C"

#endif // !RLS_FAST
