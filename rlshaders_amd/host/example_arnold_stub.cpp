// example_arnold_stub.cpp -- what an Arnold-side batching stub does with the library, without
// Arnold: gather the fields rlGgx's shader_evaluate reads from AtShaderGlobals (src/rlGgx.cpp:256-261
// of the reference) for a set of shading points, run the (sample, eval, pdf) triple for all of them
// on the GPU, read the results back.
//
// The first shading point is the probe configuration recorded in SURVEY.md section 8(c) (outputs of
// the reference's own closure code); the program prints the triple for it as JSON so that
// tests/test_gpu_host_cpp.py can compare it with tests/golden/survey_kat.json.  Exit code 0 = ok,
// 2 = no GPU (message on stderr), 1 = failure.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "rls_batch.hpp"

// the fields of Arnold's AtShaderGlobals the closures use
struct ShaderGlobalsLite { float Rd[3], N[3], Nf[3]; };

static void polar_frame(const float N[3], float U[3])
{
    // stand-in for the closed AiBuildLocalFramePolar: any unit tangent orthogonal to N will do,
    // the library takes it as an input
    float a[3] = {std::fabs(N[0]) < 0.57735f ? 1.0f : 0.0f, std::fabs(N[0]) < 0.57735f ? 0.0f : 1.0f, 0.0f};
    float c[3] = {a[1] * N[2] - a[2] * N[1], a[2] * N[0] - a[0] * N[2], a[0] * N[1] - a[1] * N[0]};
    float l = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int k = 0; k < 3; k++) U[k] = c[k] / l;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 1024;
    try {
        rlsb::Device dev(0);

        // --- the stub's per-shading-point work: append to the batch -------------------------
        rlsb::ShadingPoints pts;
        std::vector<float> rx, ry;
        {
            // point 0: N = (0,0,1), U = (1,0,0), wo = (.6, 0, .8), xi = (.25, .75)
            ShaderGlobalsLite sg = {{-0.6f, 0.0f, -0.8f}, {0, 0, 1}, {0, 0, 1}};
            float U[3] = {1, 0, 0};
            pts.add(sg.Rd, sg.N, sg.Nf, U);
            rx.push_back(0.25f); ry.push_back(0.75f);
        }
        unsigned s = 12345u;
        auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
        for (int i = 1; i < n; i++) {
            float z = 0.05f + 0.95f * rnd(), ph = 6.2831853f * rnd(), r = std::sqrt(1 - z * z);
            ShaderGlobalsLite sg = {{-r * std::cos(ph), -r * std::sin(ph), -z}, {0, 0, 1}, {0, 0, 1}};
            float U[3];
            polar_frame(sg.Nf, U);
            pts.add(sg.Rd, sg.N, sg.Nf, U);
            rx.push_back(rnd()); ry.push_back(rnd());
        }

        // --- one batched closure for all of them: specColor 1, ior 1.5, roughness sqrt(.3) ---
        rlsb::GgxSampler sampler(dev, pts, rlsb::ParamRGB(1, 1, 1), rlsb::Param(1.5f),
                                 rlsb::Param(std::sqrt(0.3f)));
        rlsb::Planes drx(dev, rx, 1), dry(dev, ry, 1);
        rlsb::Planes L(dev, n, 3), f(dev, n, 3), pdf(dev, n, 1), F(dev, n, 1);
        sampler.evalSample(drx, dry, L, F);       // GgxSampler::evalSample
        sampler.evalBrdf(L, f);                   // GgxSampler::evalBrdf
        sampler.evalPdf(L, pdf);                  // GgxSampler::evalPdf
        dev.synchronize();

        // the fused entry point must give the same bits
        rlsb::Planes L2(dev, n, 3), f2(dev, n, 3), pdf2(dev, n, 1), F2(dev, n, 1);
        sampler.sampleEvalPdf(drx, dry, L2, f2, pdf2, F2);
        std::vector<float> hL = L.download(), hf = f.download(), hp = pdf.download();
        std::vector<float> hL2 = L2.download(), hf2 = f2.download(), hp2 = pdf2.download();
        for (size_t i = 0; i < hL.size(); i++)
            if (hL[i] != hL2[i] || hf[i] != hf2[i]) { std::fprintf(stderr, "fused != separate at %zu\n", i); return 1; }
        for (size_t i = 0; i < hp.size(); i++)
            if (hp[i] != hp2[i]) { std::fprintf(stderr, "fused pdf != separate at %zu\n", i); return 1; }

        // --- the same flush recorded once and replayed as one graph launch ---------------------------------
        rlsb::Planes L3(dev, n, 3), f3(dev, n, 3), pdf3(dev, n, 1), F3(dev, n, 1);
        rlsb::Graph flush(dev, [&] {
            sampler.evalSample(drx, dry, L3, F3);
            sampler.evalBrdf(L3, f3);
            sampler.evalPdf(L3, pdf3);
        });
        flush.launch();
        dev.synchronize();
        std::vector<float> hf3 = f3.download(), hp3 = pdf3.download();
        for (size_t i = 0; i < hf.size(); i++)
            if (hf[i] != hf3[i]) { std::fprintf(stderr, "graph replay != direct at %zu\n", i); return 1; }
        for (size_t i = 0; i < hp.size(); i++)
            if (hp[i] != hp3[i]) { std::fprintf(stderr, "graph replay pdf != direct at %zu\n", i); return 1; }

        // --- batch buffers as one arena, the fastest of three candidate blocks; a direct C-ABI call on its planes ----
        rlsb::Arena arena(dev, n, 10, 3);
        {
            rls_ggx_closure c = {};
            std::vector<float> w = pts.planar(pts.wo), nn = pts.planar(pts.N), tt = pts.planar(pts.T);
            // planes of an arena are padded to 256 bytes: copy plane by plane
            for (int k = 0; k < 3; k++) {
                rlsb::check(rls_copy_to_device(dev.ctx(), arena.plane(k), w.data() + (size_t)k * n, sizeof(float) * (size_t)n));
                rlsb::check(rls_copy_to_device(dev.ctx(), arena.plane(3 + k), nn.data() + (size_t)k * n, sizeof(float) * (size_t)n));
                rlsb::check(rls_copy_to_device(dev.ctx(), arena.plane(6 + k), tt.data() + (size_t)k * n, sizeof(float) * (size_t)n));
            }
            c.wo = arena.cvec3(0); c.N = arena.cvec3(3); c.T = arena.cvec3(6);
            c.KsColor = rlsb::ParamRGB(1, 1, 1).c();
            c.specularRoughness = rlsb::Param(std::sqrt(0.3f)).c(); c.ior = rlsb::Param(1.5f).c();
            c.anisotropic = rlsb::Param(0.0f).c();
            rlsb::check(rls_ggx_pdf(dev.ctx(), n, &c, L.cvec3(), arena.plane(9)));
            std::vector<float> pa((size_t)n);
            rlsb::check(rls_copy_to_host(dev.ctx(), pa.data(), arena.plane(9), sizeof(float) * (size_t)n));
            for (int i = 0; i < n; i++)
                if (pa[(size_t)i] != hp[(size_t)i]) { std::fprintf(stderr, "arena pdf != direct at %d\n", i); return 1; }
            if (!(arena.probeGBs() > 0.0f)) { std::fprintf(stderr, "arena was not probed\n"); return 1; }
        }

        // --- rlSkin on the same shading points (defaults of src/rlSkin.cpp:109-128, sheen switched on) ---
        rlsb::SkinParams sp;
        sp.sheen_weight = rlsb::Param(0.25f);
        rlsb::SkinShader skin(dev, pts, sp);
        std::vector<float> xi6((size_t)n * 6);
        for (auto &v : xi6) v = rnd();
        rlsb::Planes dxi(dev, xi6, 6), skin_out(dev, n, rlsb::SkinShader::kOutPlanes);
        skin.sampleEvalPdf(dxi, skin_out);
        std::vector<float> hs = skin_out.download();
        const float sss_w = hs[23 * (size_t)n], spec_F = hs[22 * (size_t)n], sheen_F = hs[21 * (size_t)n];

        // --- SssSampler::integrateScatter: points on the lit plane z = 0, 16 probe rays each ------------
        std::vector<float> ns((size_t)n * 3, 0.0f), du((size_t)n * 3, 0.0f), pp((size_t)n * 3, 0.0f);
        for (int i = 0; i < n; i++) {
            ns[2 * (size_t)n + i] = 1.0f;                                  // Ns = (0,0,1)
            float ph = 6.2831853f * rnd();
            du[i] = std::cos(ph); du[(size_t)n + i] = std::sin(ph);        // dPdu in the plane
            pp[i] = rnd(); pp[(size_t)n + i] = rnd();                      // P
        }
        rlsb::Planes dNs(dev, ns, 3), dDu(dev, du, 3), dP(dev, pp, 3), scat(dev, n, 3);
        const float dist[3] = {0.05f, 0.1f, 0.2f};
        rlsb::SssSampler sss(dev, dNs, dDu, rlsb::ParamRGB(0.8f, 0.5f, 0.3f), dist);
        rls_sss_scene scene = {};
        scene.geometry = RLS_SCENE_PLANE;
        scene.plane_normal[2] = 1.0f;
        scene.light_dir[2] = 1.0f;
        scene.light_color[0] = 2.0f; scene.light_color[1] = 1.0f; scene.light_color[2] = 0.5f;
        sss.integrateScatter(dP, scene, 4, 77u, scat);
        std::vector<float> hsc = scat.download();
        double mean[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            for (int i = 0; i < n; i++) mean[k] += hsc[(size_t)k * n + i];
            mean[k] /= n;
        }

        std::printf("{\"n\": %d, \"L\": [%.9g, %.9g, %.9g], \"f\": %.9g, \"pdf\": %.9g, "
                    "\"skin\": {\"sssWeight\": %.9g, \"specularFresnel\": %.9g, \"sheenFresnel\": %.9g}, "
                    "\"scatter_mean\": [%.9g, %.9g, %.9g]}\n",
                    n, hL[0], hL[(size_t)n], hL[2 * (size_t)n], hf[0], hp[0], sss_w, spec_F, sheen_F,
                    mean[0], mean[1], mean[2]);
        return 0;
    } catch (const rlsb::Error &e) {
        std::fprintf(stderr, "rlshaders_amd: %s\n", e.what());
        return e.status == RLS_ERR_NO_DEVICE ? 2 : 1;
    }
}
