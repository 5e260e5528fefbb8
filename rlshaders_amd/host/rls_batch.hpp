// rls_batch.hpp -- C++14 host-side mirror of the reference's closure classes over the C ABI
// (include/rlshaders_amd.h).  Header-only; needs nothing but that header and the shared library.
//
// The reference builds one closure object per shading point on the stack and hands Arnold three
// static callbacks (src/rlGgx.h:97-127, src/rlGgx.cpp:261 of the reference):
//     rls::GgxSampler sampler(sg, specColor, ior, roughness, anisotropic);
//     AtVector L = GgxSampler::evalSample(&sampler, rx, ry);
//     AtColor  f = GgxSampler::evalBrdf(&sampler, &L);
//     float  pdf = GgxSampler::evalPdf(&sampler, &L);
// Here the same verbs act on a *batch* of shading points gathered by an Arnold-side stub:
//     rlsb::ShadingPoints pts;  pts.add(sg.Rd, sg.N, sg.Nf, U);  ...   // once per shading point
//     rlsb::GgxSampler sampler(dev, pts, specColor, ior, roughness, anisotropic);
//     sampler.evalSample(rx, ry, L);  sampler.evalBrdf(L, f);  sampler.evalPdf(L, pdf);
// Error behaviour follows the reference at the value level (zero vector = invalid sample, black,
// pdf 0, floors); failures of the device layer throw rlsb::Error carrying the rls_status.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "rlshaders_amd.h"

namespace rlsb {

struct Error : std::runtime_error {
    rls_status status;
    Error(rls_status s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(rls_status s)
{
    if (s != RLS_OK) {
        std::string msg = rls_last_error();
        if (msg.empty()) msg = rls_status_string(s);
        throw Error(s, msg);
    }
}

// Index-range shard of rank `rank` of `world` over `total` shading points (rls_shard_range): one Device per GPU, one
// shard each, no collective on the data path.
struct Shard { int64_t first = 0, count = 0; };
inline Shard shardRange(int64_t total, int rank, int world)
{
    Shard s;
    check(rls_shard_range(total, rank, world, &s.first, &s.count));
    return s;
}
inline int deviceCount() { return rls_device_count(); }

// Which host libm the EXACT kernels reproduce, and whether THIS process's libm is that one (rls_host_libm_matches): the
// first Device of a process asks once and says so on stderr when it is not -- results are then not bit-identical to the CPU
// closures on this host and show the alternate-libm tail of SURVEY.md Appendix D (0.009-0.14 % of chained outputs beyond 1e-5
// relative: the reference disagreeing with itself across C libraries).  RLS_QUIET_LIBM_CHECK=1 silences it.
inline const char *libmFlavour() { return rls_libm_flavour(); }
inline int hostLibmMismatches()
{
    int bad = -1;
    check(rls_host_libm_matches(&bad));
    return bad;
}
inline void warnOnceIfHostLibmDiffers()
{
    static std::once_flag once;
    std::call_once(once, [] {
        const char *quiet = std::getenv("RLS_QUIET_LIBM_CHECK");
        if (quiet && quiet[0] == '1') return;
        int bad = 0;
        if (rls_host_libm_matches(&bad) == RLS_OK && bad != 0)
            std::fprintf(stderr, "rlshaders_amd: this host's libm is not the one the library follows (%s): %d of its probe "
                                 "arguments differ -- results are within 1e-5 of this host's CPU closures, not bit-identical\n",
                         rls_libm_flavour(), bad);
    });
}

// One GPU + the stream the closures launch on.
class Device {
public:
    explicit Device(int ordinal = 0)
    {
        check(rls_context_create(ordinal, &ctx_));
        warnOnceIfHostLibmDiffers();
    }
    ~Device() { rls_context_destroy(ctx_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    rls_context *ctx() const { return ctx_; }
    void synchronize() const { check(rls_context_synchronize(ctx_)); }

private:
    rls_context *ctx_ = nullptr;
};

// A recorded flush (rls_graph): `Graph g(dev, [&] { sampler.evalSample(...); sampler.evalBrdf(...); });`
// records the calls the callable makes on `dev` instead of running them; g.launch() replays them as one
// HIP graph launch on the same buffers -- for a stub that flushes small batches every bucket.
class Graph {
public:
    template <typename Calls>
    Graph(const Device &d, Calls &&calls) : dev_(d)
    {
        check(rls_graph_begin_capture(d.ctx()));
        try {
            calls();
        } catch (...) {
            rls_graph *g = nullptr;
            if (rls_graph_end_capture(d.ctx(), &g) == RLS_OK) rls_graph_destroy(g);
            throw;
        }
        check(rls_graph_end_capture(d.ctx(), &g_));
    }
    ~Graph() { rls_graph_destroy(g_); }
    Graph(const Graph &) = delete;
    Graph &operator=(const Graph &) = delete;
    void launch() const { check(rls_graph_launch(dev_.ctx(), g_)); }

private:
    const Device &dev_;
    rls_graph *g_ = nullptr;
};

// The batch buffers of a stub: `planes` planes of n floats in ONE device allocation -- with candidates > 1 the
// fastest of that many equally sized blocks (rls_arena_create; DESIGN.md, "Placement").
class Arena {
public:
    Arena(const Device &d, int64_t n, int planes, int candidates = 1) { check(rls_arena_create(d.ctx(), n, planes, candidates, &a_)); }
    ~Arena() { rls_arena_destroy(a_); }
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    float *plane(int k) const
    {
        float *p = rls_arena_plane(a_, k);
        if (!p) throw Error(RLS_ERR_INVALID_ARGUMENT, "Arena::plane: index out of range");
        return p;
    }
    rls_cvec3 cvec3(int first) const { return rls_cvec3{plane(first), plane(first + 1), plane(first + 2)}; }
    rls_vec3 vec3(int first) const { return rls_vec3{plane(first), plane(first + 1), plane(first + 2)}; }
    rls_rgb rgb(int first) const { return rls_rgb{plane(first), plane(first + 1), plane(first + 2)}; }
    float probeGBs() const
    {
        float g = 0.0f;
        check(rls_arena_info(a_, nullptr, nullptr, &g, nullptr, nullptr));
        return g;
    }

private:
    rls_arena *a_ = nullptr;
};

// n x planes floats in device memory, planar (plane p occupies [p*n, (p+1)*n)).
class Planes {
public:
    Planes() = default;
    Planes(const Device &d, int64_t n, int planes) : dev_(&d), n_(n), planes_(planes)
    {
        void *p = nullptr;
        check(rls_device_alloc(d.ctx(), sizeof(float) * (size_t)n * (size_t)planes, &p));
        ptr_ = static_cast<float *>(p);
    }
    Planes(const Device &d, const std::vector<float> &host, int planes)
        : Planes(d, (int64_t)(host.size() / (size_t)planes), planes)
    {
        upload(host);
    }
    ~Planes() { if (ptr_) rls_device_free(dev_->ctx(), ptr_); }
    Planes(Planes &&o) noexcept { *this = std::move(o); }
    Planes &operator=(Planes &&o) noexcept
    {
        if (this != &o) {
            if (ptr_) rls_device_free(dev_->ctx(), ptr_);
            dev_ = o.dev_; ptr_ = o.ptr_; n_ = o.n_; planes_ = o.planes_;
            o.ptr_ = nullptr;
        }
        return *this;
    }
    Planes(const Planes &) = delete;
    Planes &operator=(const Planes &) = delete;

    void upload(const std::vector<float> &host)
    {
        if ((int64_t)host.size() != n_ * planes_) throw Error(RLS_ERR_INVALID_ARGUMENT, "Planes::upload: size mismatch");
        check(rls_copy_to_device(dev_->ctx(), ptr_, host.data(), sizeof(float) * host.size()));
    }
    std::vector<float> download() const
    {
        std::vector<float> host((size_t)(n_ * planes_));
        check(rls_copy_to_host(dev_->ctx(), host.data(), ptr_, sizeof(float) * host.size()));
        return host;
    }
    float *plane(int p) const { return ptr_ + (size_t)p * (size_t)n_; }
    rls_cvec3 cvec3(int first = 0) const { return rls_cvec3{plane(first), plane(first + 1), plane(first + 2)}; }
    rls_vec3 vec3(int first = 0) const { return rls_vec3{plane(first), plane(first + 1), plane(first + 2)}; }
    rls_rgb rgb(int first = 0) const { return rls_rgb{plane(first), plane(first + 1), plane(first + 2)}; }
    int64_t size() const { return n_; }
    bool empty() const { return ptr_ == nullptr; }

private:
    const Device *dev_ = nullptr;
    float *ptr_ = nullptr;
    int64_t n_ = 0;
    int planes_ = 0;
};

// n x planes floats in PAGE-LOCKED host memory (rls_host_alloc), planar: what a stub's render threads fill and read.
// Pinned memory is what lets a Pipeline's copies run asynchronously at the PCIe rate.
class HostPlanes {
public:
    HostPlanes() = default;
    HostPlanes(const Device &d, int64_t n, int planes) : dev_(&d), n_(n), planes_(planes)
    {
        void *p = nullptr;
        check(rls_host_alloc(d.ctx(), sizeof(float) * (size_t)n * (size_t)planes, &p));
        ptr_ = static_cast<float *>(p);
    }
    ~HostPlanes() { if (ptr_) rls_host_free(dev_->ctx(), ptr_); }
    HostPlanes(const HostPlanes &) = delete;
    HostPlanes &operator=(const HostPlanes &) = delete;
    float *plane(int p) const { return ptr_ + (size_t)p * (size_t)n_; }
    int64_t size() const { return n_; }
    int planes() const { return planes_; }

private:
    const Device *dev_ = nullptr;
    float *ptr_ = nullptr;
    int64_t n_ = 0;
    int planes_ = 0;
};

// A host-resident batch through the GPU in overlapped chunks (rls_pipeline_*): per chunk upload -> the closure calls `launch`
// makes -> download, on `depth` streams.  `launch(slot, first_point, count, device_in, device_out)` receives the chunk's
// rls_context (its stream) and device planes and calls the C ABI on them, e.g.
//     rlsb::Pipeline pipe(dev, 1 << 20, 19, 12);
//     pipe.run(n, in.data(), out.data(), [&](rls_context *slot, int64_t, int64_t count, float *const *i, float *const *o) {
//         rls_ggx_closure c{}; c.wo = {i[0], i[1], i[2]}; ...
//         return rls_ggx_reflect_refract(slot, count, &c, i[15], i[16], i[17], i[18], {o[0], o[1], o[2]}, ...); });
// The reference has no counterpart: it evaluates per hit on the render thread (src/rlGgx.cpp:248-261).
class Pipeline {
public:
    Pipeline(const Device &d, int64_t chunk_points, int in_planes, int out_planes, int depth = 3)
    {
        check(rls_pipeline_create(d.ctx(), chunk_points, in_planes, out_planes, depth, &p_));
    }
    ~Pipeline() { rls_pipeline_destroy(p_); }
    Pipeline(const Pipeline &) = delete;
    Pipeline &operator=(const Pipeline &) = delete;

    template <typename Launch>
    void run(int64_t n, const float *const *host_in, float *const *host_out, Launch &&launch) const
    {
        Thunk<Launch> t{&launch, nullptr};
        rls_status st = rls_pipeline_run(p_, n, host_in, host_out, &Thunk<Launch>::call, &t);
        if (t.thrown) std::rethrow_exception(t.thrown);
        check(st);
    }

private:
    template <typename Launch>
    struct Thunk {
        typename std::remove_reference<Launch>::type *fn;
        std::exception_ptr thrown;
        static rls_status call(void *user, rls_context *slot, int64_t first, int64_t count, float *const *in, float *const *out)
        {
            Thunk *t = static_cast<Thunk *>(user);
            try {
                return (*t->fn)(slot, first, count, in, out);
            } catch (...) {                  // never unwind through the C frames
                t->thrown = std::current_exception();
                return RLS_ERR_ABORTED;
            }
        }
    };
    rls_pipeline *p_ = nullptr;
};

// What the closures read from AtShaderGlobals, gathered per shading point by the Arnold-side stub:
// Rd (-> wo = -Rd), N (unflipped, only for the entering test), Nf (frame normal) and the tangent U
// that AiBuildLocalFramePolar(&U, &V, &Nf) returned (src/rlGgx.h:137-146).
struct ShadingPoints {
    std::vector<float> wo[3], N[3], T[3];
    std::vector<uint8_t> exiting;

    void add(const float Rd[3], const float Nraw[3], const float Nf[3], const float U[3])
    {
        // bool isEntering = AiV3Dot(sg->N, sg->Rd) < AI_EPSILON;   (src/rlGgx.h:137)
        const float d = Nraw[0] * Rd[0] + Nraw[1] * Rd[1] + Nraw[2] * Rd[2];
        exiting.push_back(d < 1e-4f ? 0 : 1);
        for (int k = 0; k < 3; k++) {
            wo[k].push_back(-Rd[k]);      // mViewDir = -sg->Rd  (src/rlGgx.h:144)
            N[k].push_back(Nf[k]);        // mBasis.N = mAxisN = sg->Nf  (src/rlGgx.h:145)
            T[k].push_back(U[k]);
        }
    }
    int64_t size() const { return (int64_t)wo[0].size(); }
    std::vector<float> planar(const std::vector<float> (&v)[3]) const
    {
        std::vector<float> out;
        out.reserve(v[0].size() * 3);
        for (int k = 0; k < 3; k++) out.insert(out.end(), v[k].begin(), v[k].end());
        return out;
    }
};

// A node parameter: uniform over the batch (the usual case: not texture-linked) or per point.
struct Param {
    float uniform = 0.0f;
    const float *stream = nullptr;       // device pointer, n floats
    Param(float u = 0.0f) : uniform(u) {}
    explicit Param(const float *device_plane) : stream(device_plane) {}
    rls_param c() const { return rls_param{stream, uniform}; }
};
struct ParamRGB {
    float u[3] = {1.0f, 1.0f, 1.0f};
    const float *r = nullptr, *g = nullptr, *b = nullptr;
    ParamRGB() = default;
    ParamRGB(float ur, float ug, float ub) { u[0] = ur; u[1] = ug; u[2] = ub; }
    rls_param_rgb c() const { return rls_param_rgb{r, g, b, u[0], u[1], u[2]}; }
};

// rls::GgxSampler (src/rlGgx.h:92-375), batched.
class GgxSampler {
public:
    GgxSampler(const Device &d, const ShadingPoints &sg, ParamRGB specColor, Param ior, Param roughness,
               Param anisotropic = Param(0.0f))
        : dev_(d), n_(sg.size()), wo_(d, sg.planar(sg.wo), 3), N_(d, sg.planar(sg.N), 3), T_(d, sg.planar(sg.T), 3)
    {
        c_.wo = wo_.cvec3(); c_.N = N_.cvec3(); c_.T = T_.cvec3();
        void *p = nullptr;
        check(rls_device_alloc(d.ctx(), (size_t)n_, &p));
        exiting_ = static_cast<uint8_t *>(p);
        check(rls_copy_to_device(d.ctx(), exiting_, sg.exiting.data(), (size_t)n_));
        c_.exiting = exiting_;
        c_.KsColor = specColor.c();
        c_.ior = ior.c();
        c_.specularRoughness = roughness.c();
        c_.anisotropic = anisotropic.c();
    }
    ~GgxSampler() { if (exiting_) rls_device_free(dev_.ctx(), exiting_); }
    GgxSampler(const GgxSampler &) = delete;
    GgxSampler &operator=(const GgxSampler &) = delete;

    int64_t size() const { return n_; }
    const rls_ggx_closure &closure() const { return c_; }

    // static AtVector evalSample(const void*, float rx, float ry)  (src/rlGgx.h:97-107);
    // fresnel receives what the reference adds to mReflectWeight
    void evalSample(const Planes &rx, const Planes &ry, Planes &L, Planes &fresnel) const
    {
        check(rls_ggx_sample(dev_.ctx(), n_, &c_, rx.plane(0), ry.plane(0), L.vec3(), fresnel.plane(0)));
    }
    // static AtColor evalBrdf(const void*, const AtVector *indir)  (src/rlGgx.h:110-119)
    void evalBrdf(const Planes &indir, Planes &f) const
    {
        check(rls_ggx_eval(dev_.ctx(), n_, &c_, indir.cvec3(), f.rgb()));
    }
    // static float evalPdf(const void*, const AtVector *indir)  (src/rlGgx.h:121-127)
    void evalPdf(const Planes &indir, Planes &pdf) const
    {
        check(rls_ggx_pdf(dev_.ctx(), n_, &c_, indir.cvec3(), pdf.plane(0)));
    }
    // the triple in one pass
    void sampleEvalPdf(const Planes &rx, const Planes &ry, Planes &L, Planes &f, Planes &pdf, Planes &fresnel) const
    {
        check(rls_ggx_sample_eval_pdf(dev_.ctx(), n_, &c_, rx.plane(0), ry.plane(0), L.vec3(), f.rgb(), pdf.plane(0),
                                      fresnel.plane(0)));
    }
    // per-sample body of integrateRefract  (src/rlGgx.h:228-242)
    void refractSample(const Planes &rx, const Planes &ry, Planes &dir, Planes &weight) const
    {
        check(rls_ggx_refract_sample(dev_.ctx(), n_, &c_, rx.plane(0), ry.plane(0), dir.vec3(), weight.plane(0), nullptr));
    }
    // AtColor integrateRefract(sg, data)  (src/rlGgx.h:205-245) under a uniform environment of radiance env;
    // traced = data->shouldTraceRefract(sg)
    void integrateRefract(bool traced, const float env[3], int spp_n, uint32_t seed, Planes &result,
                          uint64_t first_index = 0) const
    {
        check(rls_ggx_integrate_refract(dev_.ctx(), n_, &c_, traced ? 1 : 0, env, spp_n, seed, first_index, result.rgb(),
                                        nullptr));
    }
    // integrateGlossy's sample loop with spp_n^2 samples + getAvgReflectWeight  (src/rlGgx.h:172-184)
    // first_index: global index of this batch's point 0 (a shard draws the numbers of the whole batch)
    void integrateGlossy(int spp_n, uint32_t seed, Planes &sum_f_over_pdf, Planes &avgReflectWeight,
                         uint64_t first_index = 0) const
    {
        check(rls_ggx_integrate(dev_.ctx(), n_, &c_, spp_n, seed, first_index, sum_f_over_pdf.rgb(),
                                avgReflectWeight.plane(0)));
    }
    // the light loop of shader_evaluate (src/rlGgx.cpp:274-299) under n_lights spherical area lights:
    // direct_diffuse (Oren-Nayar, KdColor * Kd) and direct_specular (this closure, Ks)
    void directLighting(const Planes &P, const rls_sphere_light *lights, int n_lights, int spp_n, uint32_t seed,
                        ParamRGB KdColor, Param Kd, Param diffuseRoughness, Param Ks, Planes &direct_diffuse,
                        Planes &direct_specular, uint64_t first_index = 0) const
    {
        rls_ggx_shader sh{KdColor.c(), Kd.c(), diffuseRoughness.c(), Ks.c(), ParamRGB(1.0f, 1.0f, 1.0f).c(), Param(0.0f).c()};
        check(rls_ggx_direct_lighting(dev_.ctx(), n_, &c_, &sh, P.cvec3(), lights, n_lights, spp_n, seed, first_index,
                                      direct_diffuse.rgb(), direct_specular.rgb()));
    }
    // shader_evaluate of the rlGgx node for a camera ray, whole (src/rlGgx.cpp:248-327): aov = 15 planes {direct_diffuse3,
    // direct_specular3, refraction3, indirect_diffuse3, indirect_specular3}, rgb (optional) = sg->out.RGB
    void shade(const Planes &P, const rls_sphere_light *lights, int n_lights, const float env[3], bool traced, int spp_n,
               uint32_t seed, ParamRGB KdColor, Param Kd, Param diffuseRoughness, Param Ks, ParamRGB KtColor, Param Kt,
               Planes &aov, Planes *rgb = nullptr, uint64_t first_index = 0) const
    {
        rls_ggx_shader sh{KdColor.c(), Kd.c(), diffuseRoughness.c(), Ks.c(), KtColor.c(), Kt.c()};
        rls_ggx_shade_out o{};
        o.direct_diffuse = aov.rgb(0); o.direct_specular = aov.rgb(3); o.refraction = aov.rgb(6);
        o.indirect_diffuse = aov.rgb(9); o.indirect_specular = aov.rgb(12);
        if (rgb) o.out = rgb->rgb(0);
        check(rls_ggx_shade(dev_.ctx(), n_, &c_, &sh, P.cvec3(), lights, n_lights, env, traced ? 1 : 0, spp_n, seed,
                            first_index, &o));
    }
    void directLighting(const Planes &P, const rls_sphere_light &light, int spp_n, uint32_t seed, ParamRGB KdColor,
                        Param Kd, Param diffuseRoughness, Param Ks, Planes &direct_diffuse,
                        Planes &direct_specular, uint64_t first_index = 0) const
    {
        directLighting(P, &light, 1, spp_n, seed, KdColor, Kd, diffuseRoughness, Ks, direct_diffuse, direct_specular,
                       first_index);
    }

    // Parameters by reference (include/rlshaders_amd.h, rls_material_index): every streamed parameter handed to the
    // constructor is then a per-MATERIAL column of `count` floats and point i takes entry device_ids[i]
    void setMaterials(const uint32_t *device_ids, uint32_t count) { c_.materials = rls_material_index{device_ids, count}; }

private:
    const Device &dev_;
    int64_t n_;
    Planes wo_, N_, T_;
    uint8_t *exiting_ = nullptr;
    rls_ggx_closure c_{};
};

// DisneySampler (src/rlDisney.cpp:105-602), batched; parameter names from src/rlDisney.cpp:606-610.
struct DisneyParams {
    ParamRGB base_color;
    Param subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint, clearcoat,
        clearcoat_gloss;
};

class DisneySampler {
public:
    DisneySampler(const Device &d, const ShadingPoints &sg, const DisneyParams &p)
        : dev_(d), n_(sg.size()), wo_(d, sg.planar(sg.wo), 3), N_(d, sg.planar(sg.N), 3), T_(d, sg.planar(sg.T), 3)
    {
        c_.wo = wo_.cvec3(); c_.N = N_.cvec3(); c_.T = T_.cvec3();
        c_.base_color = p.base_color.c();
        c_.subsurface = p.subsurface.c(); c_.metallic = p.metallic.c(); c_.specular = p.specular.c();
        c_.specular_tint = p.specular_tint.c(); c_.roughness = p.roughness.c(); c_.anisotropic = p.anisotropic.c();
        c_.sheen = p.sheen.c(); c_.sheen_tint = p.sheen_tint.c(); c_.clearcoat = p.clearcoat.c();
        c_.clearcoat_gloss = p.clearcoat_gloss.c();
    }
    // inline void setSampleType(AtUInt16 type)  (src/rlDisney.cpp:194-197)
    void setSampleType(int ray_type)
    {
        if (ray_type != RLS_RAY_DIFFUSE && ray_type != RLS_RAY_GLOSSY)
            throw Error(RLS_ERR_INVALID_ARGUMENT, "setSampleType: RLS_RAY_DIFFUSE or RLS_RAY_GLOSSY");
        mSampleType = ray_type;
    }
    void evalSample(const Planes &rx, const Planes &ry, Planes &L) const
    {
        check(rls_disney_sample(dev_.ctx(), n_, &c_, mSampleType, rx.plane(0), ry.plane(0), L.vec3()));
    }
    void evalBrdf(const Planes &indir, Planes &f) const
    {
        check(rls_disney_eval(dev_.ctx(), n_, &c_, mSampleType, indir.cvec3(), f.rgb()));
    }
    void evalPdf(const Planes &indir, Planes &pdf) const
    {
        check(rls_disney_pdf(dev_.ctx(), n_, &c_, mSampleType, indir.cvec3(), pdf.plane(0)));
    }
    // integrateDiffuse + integrateGlossy sample loops (src/rlDisney.cpp:240-315), spp_n^2 samples per lobe:
    // per point and lobe the sum of eval/pdf over the valid samples and their count
    void integrate(int spp_n, uint32_t seed, Planes &diffuse_sum, Planes &diffuse_count, Planes &specular_sum,
                   Planes &specular_count, uint64_t first_index = 0) const
    {
        check(rls_disney_integrate(dev_.ctx(), n_, &c_, spp_n, seed, first_index, diffuse_sum.rgb(), diffuse_count.plane(0),
                                   specular_sum.rgb(), specular_count.plane(0), nullptr));
    }
    // the light loop of shader_evaluate (src/rlDisney.cpp:695-705): evalDiffuseLightSample + evalSpecularLightSample
    // per light -> the two direct AOVs
    void directLighting(const Planes &P, const rls_sphere_light *lights, int n_lights, int spp_n, uint32_t seed,
                        Planes &direct_diffuse, Planes &direct_specular, uint64_t first_index = 0) const
    {
        check(rls_disney_direct_lighting(dev_.ctx(), n_, &c_, P.cvec3(), lights, n_lights, spp_n, seed, first_index,
                                         direct_diffuse.rgb(), direct_specular.rgb()));
    }
    // shader_evaluate of the rlDisney node for a camera ray, whole (src/rlDisney.cpp:685-727): aov = 12 planes
    // {direct_diffuse3, direct_specular3, indirect_diffuse3, indirect_specular3}, rgb (optional) = sg->out.RGB
    void shade(const Planes &P, const rls_sphere_light *lights, int n_lights, const float env[3], int spp_n, uint32_t seed,
               Planes &aov, Planes *rgb = nullptr, uint64_t first_index = 0) const
    {
        rls_disney_shade_out o{};
        o.direct_diffuse = aov.rgb(0); o.direct_specular = aov.rgb(3); o.indirect_diffuse = aov.rgb(6);
        o.indirect_specular = aov.rgb(9);
        if (rgb) o.out = rgb->rgb(0);
        check(rls_disney_shade(dev_.ctx(), n_, &c_, P.cvec3(), lights, n_lights, env, spp_n, seed, first_index, &o));
    }
    // the same with every sample handed to `consume` chunk by chunk (the loop body of src/rlDisney.cpp:299-312);
    // chunk_wi / chunk_f: 3 x (2 * spp * chunk_points) planes, chunk_pdf: 1 x the same
    void integrateStreamed(int spp_n, uint32_t seed, Planes &diffuse_sum, Planes &diffuse_count, Planes &specular_sum,
                           Planes &specular_count, int64_t chunk_points, Planes &chunk_wi, Planes &chunk_f,
                           Planes &chunk_pdf, rls_disney_chunk_fn consume, void *user, uint64_t first_index = 0) const
    {
        rls_disney_stream_out so{chunk_wi.vec3(), chunk_f.rgb(), chunk_pdf.plane(0)};
        check(rls_disney_integrate_chunked(dev_.ctx(), n_, &c_, spp_n, seed, first_index, diffuse_sum.rgb(),
                                           diffuse_count.plane(0), specular_sum.rgb(), specular_count.plane(0),
                                           chunk_points, &so, consume, user));
    }
    int64_t size() const { return n_; }

    // Parameters by reference (include/rlshaders_amd.h, rls_material_index): every streamed parameter handed to the
    // constructor is then a per-MATERIAL column of `count` floats and point i takes entry device_ids[i]
    void setMaterials(const uint32_t *device_ids, uint32_t count) { c_.materials = rls_material_index{device_ids, count}; }

private:
    const Device &dev_;
    int64_t n_;
    Planes wo_, N_, T_;
    rls_disney_closure c_{};
    int mSampleType = RLS_RAY_GLOSSY;
};

// rls::NDProfile (src/rlSss.h:27-61), batched with uniform or streamed distances.
class NDProfile {
public:
    NDProfile(const Device &d, int64_t n) : dev_(d), n_(n)
    {
        c_.sss_color = ParamRGB().c();
        c_.sss_dist_multiplier = Param(1.0f).c();
        for (auto &x : c_.sss_scatter_dist) x = Param(1.0f).c();
    }
    // void setDistance(const AtVector &dist, const AtColor &albedo)  (src/rlSss.cpp:20-34)
    void setDistance(const float dist[3], const float albedo[3])
    {
        for (int k = 0; k < 3; k++) c_.sss_scatter_dist[k] = Param(dist[k]).c();
        c_.sss_color = ParamRGB(albedo[0], albedo[1], albedo[2]).c();
    }
    // getRadius / getPdf / evalProfile  (src/rlSss.cpp:36-106)
    void sample(const Planes &rx, Planes &r, Planes &pdf, Planes &profile) const
    {
        check(rls_nd_sample(dev_.ctx(), n_, &c_, rx.plane(0), r.plane(0), pdf.plane(0), profile.rgb()));
    }
    void getPdf(const Planes &r, Planes &pdf) const { check(rls_nd_pdf(dev_.ctx(), n_, &c_, r.plane(0), pdf.plane(0))); }
    void evalProfile(const Planes &r, Planes &profile) const
    {
        check(rls_nd_eval(dev_.ctx(), n_, &c_, r.plane(0), profile.rgb()));
    }

    // Parameters by reference (include/rlshaders_amd.h, rls_material_index): every streamed parameter handed to the
    // constructor is then a per-MATERIAL column of `count` floats and point i takes entry device_ids[i]
    void setMaterials(const uint32_t *device_ids, uint32_t count) { c_.materials = rls_material_index{device_ids, count}; }

private:
    const Device &dev_;
    int64_t n_;
    rls_sss_closure c_{};
};

// rls::SssSampler<NDProfile> (src/rlSss.h:100-560), batched.  Ns = sg->Ns and dPdu = sg->dPdu per
// shading point (the constructor's frame, 143-158), P = sg->P.
class SssSampler {
public:
    SssSampler(const Device &d, const Planes &Ns, const Planes &dPdu, ParamRGB albedo, const float dist[3],
               Param multiplier = Param(1.0f), bool has_dPdu = true)
        : dev_(d), n_(Ns.size())
    {
        c_.sss_color = albedo.c();
        c_.sss_dist_multiplier = multiplier.c();
        for (int k = 0; k < 3; k++) c_.sss_scatter_dist[k] = Param(dist[k]).c();
        c_.N = Ns.cvec3();
        c_.T = dPdu.cvec3();
        c_.has_dPdu = has_dPdu ? 1 : 0;
    }
    // float getProbeRay(rx, ry, origin, ray)  (src/rlSss.h:487-533), plus pdf(r) and profile(r)
    void getProbeRay(const Planes &rx, const Planes &ry, const Planes &P, Planes &r, Planes &origin, Planes &dir,
                     Planes &maxdist, Planes &pdf, Planes &profile) const
    {
        check(rls_sss_probe_ray(dev_.ctx(), n_, &c_, rx.plane(0), ry.plane(0), P.cvec3(), r.plane(0), origin.vec3(),
                                dir.vec3(), maxdist.plane(0), pdf.plane(0), profile.rgb()));
    }
    // the MIS pdf of one probe hit (src/rlSss.h:246-266)
    void misPdf(const Planes &disp, const Planes &sampleN, Planes &pdf, bool literal_matrix = false) const
    {
        check(rls_sss_mis_pdf(dev_.ctx(), n_, &c_, disp.cvec3(), sampleN.cvec3(), literal_matrix ? 1 : 0, pdf.plane(0)));
    }
    // AtColor integrateScatter(sg, data)  (src/rlSss.h:167-280) over an analytic scene
    void integrateScatter(const Planes &P, const rls_sss_scene &scene, int spp_n, uint32_t seed, Planes &result,
                          uint64_t first_index = 0) const
    {
        check(rls_sss_integrate_scatter(dev_.ctx(), n_, &c_, P.cvec3(), &scene, spp_n, seed, first_index, result.rgb(),
                                        nullptr));
    }

    // Parameters by reference (include/rlshaders_amd.h, rls_material_index): every streamed parameter handed to the
    // constructor is then a per-MATERIAL column of `count` floats and point i takes entry device_ids[i]
    void setMaterials(const uint32_t *device_ids, uint32_t count) { c_.materials = rls_material_index{device_ids, count}; }

private:
    const Device &dev_;
    int64_t n_;
    rls_sss_closure c_{};
};

// rlSkin's lobe composition (shader_evaluate, src/rlSkin.cpp:174-246), batched; parameter names and
// defaults of node_parameters (src/rlSkin.cpp:109-128).
struct SkinParams {
    ParamRGB sss_color{1.0f, 1.0f, 1.0f};
    Param sss_weight{1.0f}, sss_dist_multiplier{1.0f};
    Param sss_scatter_dist[3] = {Param(1.0f), Param(1.0f), Param(1.0f)};
    ParamRGB specular_color{1.0f, 1.0f, 1.0f};
    Param specular_weight{0.6f}, specular_roughness{0.5f}, specular_ior{1.44f};
    ParamRGB sheen_color{1.0f, 1.0f, 1.0f};
    Param sheen_weight{0.0f}, sheen_roughness{0.35f}, sheen_ior{1.44f};
};

class SkinShader {
public:
    SkinShader(const Device &d, const ShadingPoints &sg, const SkinParams &p)
        : dev_(d), n_(sg.size()), wo_(d, sg.planar(sg.wo), 3), N_(d, sg.planar(sg.N), 3), T_(d, sg.planar(sg.T), 3)
    {
        c_.wo = wo_.cvec3(); c_.N = N_.cvec3(); c_.T = T_.cvec3();
        c_.sss_color = p.sss_color.c(); c_.sss_weight = p.sss_weight.c();
        c_.sss_dist_multiplier = p.sss_dist_multiplier.c();
        for (int k = 0; k < 3; k++) c_.sss_scatter_dist[k] = p.sss_scatter_dist[k].c();
        c_.specular_color = p.specular_color.c(); c_.specular_weight = p.specular_weight.c();
        c_.specular_roughness = p.specular_roughness.c(); c_.specular_ior = p.specular_ior.c();
        c_.sheen_color = p.sheen_color.c(); c_.sheen_weight = p.sheen_weight.c();
        c_.sheen_roughness = p.sheen_roughness.c(); c_.sheen_ior = p.sheen_ior.c();
    }
    // xi: 6 planes {sheen rx, ry, specular rx, ry, sss rx, ry};  out: 27 planes in rls_skin_out order
    // (sheen wi3 f3 pdf fresnel, specular wi3 f3 pdf fresnel, r, r_pdf, profile3, sheenFresnel,
    // specularFresnel, sssWeight)
    void sampleEvalPdf(const Planes &xi, Planes &out) const
    {
        const float *x[6];
        for (int k = 0; k < 6; k++) x[k] = xi.plane(k);
        rls_skin_out o{};
        o.sheen_wi = out.vec3(0); o.sheen_f = out.rgb(3); o.sheen_pdf = out.plane(6); o.sheen_fresnel = out.plane(7);
        o.spec_wi = out.vec3(8);  o.spec_f = out.rgb(11); o.spec_pdf = out.plane(14); o.spec_fresnel = out.plane(15);
        o.r = out.plane(16); o.r_pdf = out.plane(17); o.profile = out.rgb(18);
        o.sheenFresnel = out.plane(21); o.specularFresnel = out.plane(22); o.sssWeight = out.plane(23);
        check(rls_skin_sample_eval_pdf(dev_.ctx(), n_, &c_, x, &o));
    }
    static constexpr int kOutPlanes = 24;
    // shader_evaluate over spp_n^2 samples per layer  (src/rlSkin.cpp:174-254): the mean Fresnel of each GGX lobe
    // (getAvgReflectWeight, src/rlGgx.h:181-184) handed down to the next, integrateScatter x sssWeight.
    // aov: 12 planes {sheen3, specular3, sss3, out3}; layers (optional): 3 planes {sheenFresnel, specularFresnel, sssWeight}
    // lights / n_lights: the spherical lights of the two light loops (193-198, 217-222); none by default
    void integrate(const Planes &P, const rls_sss_scene &scene, const float env[3], int spp_n, uint32_t seed, Planes &aov,
                   Planes *layers = nullptr, uint64_t first_index = 0, const rls_sphere_light *lights = nullptr,
                   int n_lights = 0) const
    {
        rls_skin_integrate_out o{};
        o.sheen = aov.rgb(0); o.specular = aov.rgb(3); o.sss = aov.rgb(6); o.out = aov.rgb(9);
        if (layers) { o.sheenFresnel = layers->plane(0); o.specularFresnel = layers->plane(1); o.sssWeight = layers->plane(2); }
        check(rls_skin_integrate(dev_.ctx(), n_, &c_, P.cvec3(), &scene, env, lights, n_lights, spp_n, seed, first_index, &o));
    }

    // Parameters by reference (include/rlshaders_amd.h, rls_material_index): every streamed parameter handed to the
    // constructor is then a per-MATERIAL column of `count` floats and point i takes entry device_ids[i]
    void setMaterials(const uint32_t *device_ids, uint32_t count) { c_.materials = rls_material_index{device_ids, count}; }

private:
    const Device &dev_;
    int64_t n_;
    Planes wo_, N_, T_;
    rls_skin_closure c_{};
};

} // namespace rlsb
