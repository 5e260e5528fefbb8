// example_host_pipeline.cpp -- a host-resident batch through rlsb::Pipeline: the shading points live in page-locked HOST
// memory (where an Arnold-side stub's render threads put them; the reference evaluates per hit on those threads,
// src/rlGgx.cpp:248-261), travel through the GPU in chunks on three streams -- upload, rls_ggx_reflect_refract, download,
// overlapped -- and the results land in host memory.
//
//   example_host_pipeline <log2 points> <log2 chunk points> <depth> [passes] [shade]
// prints one JSON line: the pipeline's throughput, the box's pinned copy rates to hold it against, and whether the
// pipeline's outputs equal those of ONE device-resident call on the same points bit for bit (exit code 1 if not).
// With `shade` the unit that crosses the bus is a shading point instead of a sample: rls_ggx_shade (the whole
// shader_evaluate of rlGgx, 144 samples per point) per chunk, its sixteen node parameters BY REFERENCE -- a material id per
// point, 256 node instances' parameters as device-resident columns (rls_material_index) -- so that 52 B go up (wo3 N3 T3 P3
// id) and 12 B come down (sg->out.RGB; the AOV planes are written in the slot and not downloaded) per shading point.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rls_batch.hpp"

namespace {
constexpr uint32_t kSeed = 1234;
constexpr int kIn = 19, kOut = 12;      // wo3 N3 T3 Ks3 rough ior aniso xi4 -> wi3 f3 pdf F wt3 weight

rls_status launch(rls_context *slot, int64_t, int64_t count, float *const *i, float *const *o)
{
    rls_ggx_closure c = {};
    c.wo = rls_cvec3{i[0], i[1], i[2]}; c.N = rls_cvec3{i[3], i[4], i[5]}; c.T = rls_cvec3{i[6], i[7], i[8]};
    c.KsColor = rls_param_rgb{i[9], i[10], i[11], 0, 0, 0};
    c.specularRoughness = rls_param{i[12], 0}; c.ior = rls_param{i[13], 0}; c.anisotropic = rls_param{i[14], 0};
    return rls_ggx_reflect_refract(slot, count, &c, i[15], i[16], i[17], i[18], rls_vec3{o[0], o[1], o[2]},
                                   rls_rgb{o[3], o[4], o[5]}, o[6], o[7], rls_vec3{o[8], o[9], o[10]}, o[11]);
}

// ---- whole-node mode -----------------------------------------------------------------------------------------------
constexpr int kShadeIn = 13, kShadeOut = 18, kMaterials = 256, kColumns = 16;
struct ShadeScene {
    const float *col[kColumns];          // Ks3 rough ior aniso KdColor3 Kd KdRough Ks KtColor3 Kt, kMaterials entries each
    rls_sphere_light lights[2];
    float env[3];
};
ShadeScene g_scene;

rls_status launch_shade(rls_context *slot, int64_t first, int64_t count, float *const *i, float *const *o)
{
    const float *const *t = g_scene.col;
    rls_ggx_closure c = {};
    c.wo = rls_cvec3{i[0], i[1], i[2]}; c.N = rls_cvec3{i[3], i[4], i[5]}; c.T = rls_cvec3{i[6], i[7], i[8]};
    c.KsColor = rls_param_rgb{t[0], t[1], t[2], 0, 0, 0};
    c.specularRoughness = rls_param{t[3], 0}; c.ior = rls_param{t[4], 0}; c.anisotropic = rls_param{t[5], 0};
    c.materials = rls_material_index{reinterpret_cast<const uint32_t *>(i[12]), (uint32_t)kMaterials};
    rls_ggx_shader sh = {};
    sh.KdColor = rls_param_rgb{t[6], t[7], t[8], 0, 0, 0};
    sh.Kd = rls_param{t[9], 0}; sh.diffuseRoughness = rls_param{t[10], 0}; sh.Ks = rls_param{t[11], 0};
    sh.KtColor = rls_param_rgb{t[12], t[13], t[14], 0, 0, 0};
    sh.Kt = rls_param{t[15], 0};
    rls_ggx_shade_out out = {};
    out.direct_diffuse = rls_rgb{o[0], o[1], o[2]}; out.direct_specular = rls_rgb{o[3], o[4], o[5]};
    out.refraction = rls_rgb{o[6], o[7], o[8]}; out.indirect_diffuse = rls_rgb{o[9], o[10], o[11]};
    out.indirect_specular = rls_rgb{o[12], o[13], o[14]}; out.out = rls_rgb{o[15], o[16], o[17]};
    // first_index = the chunk's first point: every point draws the sample numbers it has in the whole batch
    return rls_ggx_shade(slot, count, &c, &sh, rls_cvec3{i[9], i[10], i[11]}, g_scene.lights, 2, g_scene.env, 1, 4, kSeed,
                         (uint64_t)first, &out);
}

int shade_mode(int log2n, int log2c, int depth, int passes)
{
    const int64_t n = ((int64_t)1 << log2n) - 37;
    rlsb::Device dev(0);
    rls_context *ctx = dev.ctx();
    rlsb::Planes din(dev, n, kShadeIn), dref(dev, n, kShadeOut), table(dev, kMaterials, kColumns);
    rlsb::check(rls_gen_frame(ctx, kSeed, 0, n, din.vec3(0), din.vec3(3), din.vec3(6)));
    for (int j = 0; j < 3; j++) rlsb::check(rls_gen_uniform(ctx, kSeed, 0, n, 40 + j, 0.0f, 4.0f, din.plane(9 + j)));   // P
    // the material id of every point: floor(u * 256) computed on the host from a device-generated uniform plane
    {
        rlsb::Planes u(dev, n, 1);
        rlsb::check(rls_gen_uniform(ctx, kSeed, 0, n, 62, 0.0f, 1.0f, u.plane(0)));
        std::vector<float> h = u.download();
        std::vector<float> bits(h.size());
        for (size_t k = 0; k < h.size(); k++) {
            uint32_t id = (uint32_t)(h[k] * (float)kMaterials);
            if (id >= (uint32_t)kMaterials) id = kMaterials - 1;
            std::memcpy(&bits[k], &id, sizeof id);
        }
        rlsb::check(rls_copy_to_device(ctx, din.plane(12), bits.data(), sizeof(float) * bits.size()));
    }
    const float lo[kColumns] = {0, 0, 0, 0.05f, 1.05f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float hi[kColumns] = {1, 1, 1, 1.0f, 2.55f, 0.9f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    for (int k = 0; k < kColumns; k++) {
        rlsb::check(rls_gen_uniform(ctx, kSeed, 0, kMaterials, 8 + k, lo[k], hi[k], table.plane(k)));
        g_scene.col[k] = table.plane(k);
    }
    g_scene.lights[0] = rls_sphere_light{{2.0f, 2.0f, 6.0f}, 1.25f, {3.0f, 2.0f, 1.0f}, RLS_MIS_BOTH};
    g_scene.lights[1] = rls_sphere_light{{-1.0f, 5.0f, 7.0f}, 0.6f, {0.5f, 0.5f, 4.0f}, RLS_MIS_BOTH};
    g_scene.env[0] = 1.0f; g_scene.env[1] = 0.9f; g_scene.env[2] = 0.8f;

    rlsb::HostPlanes hin(dev, n, kShadeIn), hout(dev, n, 3);
    rlsb::check(rls_copy_to_host(ctx, hin.plane(0), din.plane(0), sizeof(float) * (size_t)n * kShadeIn));
    float *di[kShadeIn], *dr[kShadeOut];
    for (int k = 0; k < kShadeIn; k++) di[k] = din.plane(k);
    for (int k = 0; k < kShadeOut; k++) dr[k] = dref.plane(k);
    rlsb::check(launch_shade(ctx, 0, n, di, dr));                         // reference: one device-resident call
    std::vector<float> ref = dref.download();

    std::vector<const float *> in(kShadeIn);
    std::vector<float *> out(kShadeOut, nullptr);                          // the five AOVs: not downloaded
    for (int k = 0; k < kShadeIn; k++) in[k] = hin.plane(k);
    for (int k = 0; k < 3; k++) out[15 + k] = hout.plane(k);
    rlsb::Pipeline pipe(dev, (int64_t)1 << log2c, kShadeIn, kShadeOut, depth);
    std::memset(hout.plane(0), 0xff, sizeof(float) * (size_t)n * 3);
    pipe.run(n, in.data(), out.data(), launch_shade);
    const bool same = std::memcmp(hout.plane(0), ref.data() + (size_t)15 * (size_t)n, sizeof(float) * (size_t)n * 3) == 0;
    double best = 1e30;
    for (int p = 0; p < passes; p++) {
        const auto t0 = std::chrono::steady_clock::now();
        pipe.run(n, in.data(), out.data(), launch_shade);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (s < best) best = s;
    }
    float rates[3] = {0, 0, 0};
    rlsb::check(rls_measure_copy_rates(ctx, (size_t)256 << 20, rates));
    std::printf("{\"mode\": \"shade\", \"points\": %lld, \"chunk_points\": %lld, \"depth\": %d, \"seconds\": %.6f, "
                "\"gsamples_per_s\": %.4f, \"h2d_gb_per_s\": %.2f, \"d2h_gb_per_s\": %.2f, \"box_h2d\": %.2f, \"box_d2h\": %.2f, "
                "\"bit_identical_to_device_resident\": %s}\n",
                (long long)n, (long long)1 << log2c, depth, best, 144.0 * (double)n / best / 1e9, 4.0 * kShadeIn * (double)n / best / 1e9,
                12.0 * (double)n / best / 1e9, rates[0], rates[1], same ? "true" : "false");
    return same ? 0 : 1;
}
} // namespace

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? std::atoi(argv[1]) : 22;
    const int log2c = argc > 2 ? std::atoi(argv[2]) : 18;
    const int depth = argc > 3 ? std::atoi(argv[3]) : 3;
    const int passes = argc > 4 ? std::atoi(argv[4]) : 3;
    const int64_t n = ((int64_t)1 << log2n) - 37;           // ragged: the last chunk is short
    try {
        if (argc > 5 && std::strcmp(argv[5], "shade") == 0) return shade_mode(log2n, log2c, depth, passes);
        rlsb::Device dev(0);
        rls_context *ctx = dev.ctx();
        // the synthetic batch, generated on the device once and moved to pinned host memory: from here on it is "host data"
        rlsb::Planes din(dev, n, kIn), dref(dev, n, kOut);
        rlsb::check(rls_gen_frame(ctx, kSeed, 0, n, din.vec3(0), din.vec3(3), din.vec3(6)));
        for (int j = 0; j < 3; j++) rlsb::check(rls_gen_uniform(ctx, kSeed, 0, n, 8 + j, 0.0f, 1.0f, din.plane(9 + j)));
        rlsb::check(rls_gen_uniform(ctx, kSeed, 0, n, 5, 0.05f, 1.0f, din.plane(12)));
        rlsb::check(rls_gen_uniform(ctx, kSeed, 0, n, 6, 1.05f, 2.55f, din.plane(13)));
        rlsb::check(rls_gen_aniso(ctx, kSeed, 0, n, din.plane(14)));
        for (int j = 0; j < 4; j++) rlsb::check(rls_gen_uniform(ctx, kSeed, 0, n, 11 + j, 0.0f, 1.0f, din.plane(15 + j)));
        rlsb::HostPlanes hin(dev, n, kIn), hout(dev, n, kOut);
        rlsb::check(rls_copy_to_host(ctx, hin.plane(0), din.plane(0), sizeof(float) * (size_t)n * kIn));
        // reference: one device-resident call over the whole batch
        float *di[kIn], *dr[kOut];
        for (int k = 0; k < kIn; k++) di[k] = din.plane(k);
        for (int k = 0; k < kOut; k++) dr[k] = dref.plane(k);
        rlsb::check(launch(ctx, 0, n, di, dr));
        std::vector<float> ref = dref.download();

        std::vector<const float *> in(kIn);
        std::vector<float *> out(kOut);
        for (int k = 0; k < kIn; k++) in[k] = hin.plane(k);
        for (int k = 0; k < kOut; k++) out[k] = hout.plane(k);
        rlsb::Pipeline pipe(dev, (int64_t)1 << log2c, kIn, kOut, depth);
        std::memset(hout.plane(0), 0xff, sizeof(float) * (size_t)n * kOut);
        pipe.run(n, in.data(), out.data(), launch);                       // warm-up, and the pass that is checked
        const bool same = std::memcmp(hout.plane(0), ref.data(), sizeof(float) * (size_t)n * kOut) == 0;
        double best = 1e30;
        for (int p = 0; p < passes; p++) {
            const auto t0 = std::chrono::steady_clock::now();
            pipe.run(n, in.data(), out.data(), launch);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s < best) best = s;
        }
        float rates[3] = {0, 0, 0};
        rlsb::check(rls_measure_copy_rates(ctx, (size_t)256 << 20, rates));
        const double up = 4.0 * kIn * (double)n, down = 4.0 * kOut * (double)n;
        std::printf("{\"points\": %lld, \"chunk_points\": %lld, \"depth\": %d, \"seconds\": %.6f, \"gsamples_per_s\": %.4f, "
                    "\"h2d_gb_per_s\": %.2f, \"d2h_gb_per_s\": %.2f, \"box_h2d\": %.2f, \"box_d2h\": %.2f, \"box_both\": %.2f, "
                    "\"bit_identical_to_device_resident\": %s}\n",
                    (long long)n, (long long)1 << log2c, depth, best, 2.0 * (double)n / best / 1e9, up / best / 1e9,
                    down / best / 1e9, rates[0], rates[1], rates[2], same ? "true" : "false");
        return same ? 0 : 1;
    } catch (const rlsb::Error &e) {
        std::fprintf(stderr, "rlshaders_amd: %s\n", e.what());
        return e.status == RLS_ERR_NO_DEVICE ? 2 : 1;
    }
}
