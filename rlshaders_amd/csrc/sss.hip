// sss.hip -- rlSss kernels: NDProfile (src/rlSss.h:27-61, src/rlSss.cpp:20-106 of the reference)
// and the hot parts of SssSampler<NDProfile> (src/rlSss.h:143-167,246-266,401-413,487-545), plus
// the rlUtil direction helpers, with their C-ABI entry points.  gfx950, one point per lane.
//
// Roofline: HBM.  Algorithmic bytes per sample: probe ray 56 B in (dist3 albedo3 N3 T3 xi2) +
// 48 B out (r, origin3, dir3, maxdist, pdf, R3) = 104 B; profile-only 32 B in + 20 B out = 52 B
// (SURVEY.md section 8(d), config 4).
// loads behind reload_args / the per-parameter stream-or-uniform branches sit in later basic blocks than make_idx():
// they renew the lane-offset barrier (rls_device.hpp) so that every plane access keeps the scalar-base addressing form
#ifndef RLS_LOAD_RENEW
#define RLS_LOAD_RENEW 1
#endif
#include "rls_internal.hpp"

using namespace rlsd;

namespace {

using rlsh::SssIO;
using rlsh::MiscIO;
enum { OP_ND = rlsh::SOP_ND, OP_ND_PDF = rlsh::SOP_ND_PDF, OP_ND_EVAL = rlsh::SOP_ND_EVAL, OP_PROBE = rlsh::SOP_PROBE,
       OP_MIS = rlsh::SOP_MIS };
enum { OP_CAVITY = rlsh::MOP_CAVITY, OP_DIFFUSE_DIR = rlsh::MOP_DIFFUSE_DIR, OP_UTIL = rlsh::MOP_UTIL,
       OP_REFLECT_LUM = rlsh::MOP_REFLECT_LUM };

// (the one-sample kernels keep the reciprocals of d_i only: those of c1 + 3 c2 would serve one division each -- measured, not kept)
// INDEXED: the parameters are per-material columns (rls_material_index); a specialisation, so that the kernels launched
// without a table keep their code
template <bool INDEXED, class I>
__device__ __forceinline__ NdProfile load_profile(const rls_sss_closure &c, I i)
{
    // scatterDist = sss_scatter_dist * sss_dist_multiplier (src/rlSkin.cpp:235-236)
    const PIndex<I> k = pindex<INDEXED>(c.materials, i);
    float m = ldp(c.sss_dist_multiplier, k);
    float dx = ldp(c.sss_scatter_dist[0], k) * m;
    float dy = ldp(c.sss_scatter_dist[1], k) * m;
    float dz = ldp(c.sss_scatter_dist[2], k) * m;
    return nd_make<false>(dx, dy, dz);
}

// UNIFORM: the scatter distance and its multiplier are one value for the batch (an Arnold parameter is a constant unless a
// texture is linked to it): setDistance -- three divisions, six expf, with all of getPdf's reciprocals -- runs once per
// thread ahead of the tile loop, the same values a per-point evaluation gives
__device__ __forceinline__ NdProfile uniform_profile(const rls_sss_closure &c)
{
    const float m = c.sss_dist_multiplier.u;
    const NdProfile p = nd_make<!RLS_FAST>(c.sss_scatter_dist[0].u * m, c.sss_scatter_dist[1].u * m, c.sss_scatter_dist[2].u * m);
    // measured (tools/ab.sh, probe ray at 2^26 points): 1.258 ms from scalar registers, 1.240 from vector registers -- these
    // kernels have the vector registers to spare and every use of a scalar operand beyond the first costs a move
    // (an explicit copy: plain `return p;` makes the compiler order the same instructions differently; this form keeps the
    // kernels byte-identical to the ones the committed counter profiles were taken on -- rlshaders_amd/codeid.py)
    return NdProfile(p);
}

enum { PER_POINT = 0, UNIFORM_DISTANCE = 1, BY_REFERENCE = 2 };
// occupancy (waves per SIMD the register allocator must allow).  Left alone the kernels take 62 vector and 106 scalar registers:
// seven waves.  Pinned at eight (78 scalar registers, nothing spilled) the probe runs 3.2 % faster (1.479 -> 1.432 ms) and NDProfile
// alone 1.9 %; with a uniform scatter distance eight LOSES 2.7 % and seven is what the compiler picks; five +3 %, six 0
// (tools/ab.sh, two interleaved repetitions).
#ifndef RLS_SSS_WAVES
#define RLS_SSS_WAVES(MODE) ((MODE) == UNIFORM_DISTANCE ? 7 : 8)
#endif
#define RLS_SSS_ATTR(MODE) __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_SSS_WAVES(MODE), RLS_SSS_WAVES(MODE))))
// the kernel body: inlined into sss_kernel (the product) and sss_kernel_stamped (diagnostic: the same body between clock
// stamps, rls_internal.hpp ClockStamp).  a0 is the kernel's first parameter (reload_args).
template <int OP, int MODE, int FAST_MATH>
__device__ __forceinline__ void sss_body(const SssIO &a0)
{
    stage_libm_tables();   // expf / logf tables -> LDS (EXACT mode)
    constexpr bool UNIFORM = MODE == UNIFORM_DISTANCE;
    NdProfile pu = {};
    if (UNIFORM) pu = uniform_profile(a0.c);
    const TileRange tiles = tile_range(a0.n);
    // (two tiles per loop iteration and lane were tried in round 4 and lose: profiles/r04_sss_two_points.txt)
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const Idx i = make_idx(base);
        if (i.full() >= a0.n) continue;
        const SssIO a = reload_args(a0);       // plane pointers re-read per tile (rls_internal.hpp, reload_args)
        NdProfile p = UNIFORM ? pu : load_profile<MODE == BY_REFERENCE>(a.c, i);
        if (OP == OP_ND) {
            float r = nd_radius(p, ldg(a.rx, i));
            float pdf, R, G, B;
            nd_pdf_profile(p, r, pdf, R, G, B);
            stg(a.r, i, r);
            stg(a.pdf, i, pdf);
            strgb(a.profile, i, R, G, B);
        } else if (OP == OP_ND_PDF) {
            stg(a.pdf, i, nd_pdf(p, ldg(a.rin, i)));
        } else if (OP == OP_ND_EVAL) {
            float R, G, B;
            nd_profile(p, ldg(a.rin, i), R, G, B);
            strgb(a.profile, i, R, G, B);
        } else if (OP == OP_PROBE) {
            Frame fr = sss_frame(ld3(a.c.N, i), ld3(a.c.T, i), a.c.has_dPdu != 0);
            V3 off, dir;
            float maxdist;
            float r = sss_probe_ray(p, fr, ldg(a.rx, i), ldg(a.ry, i), off, dir, maxdist);
            if (a.P.x) off = ld3(a.P, i) + off;                       // ray.origin = origin + offset
            float pdf, R, G, B;
            nd_pdf_profile(p, r, pdf, R, G, B);
            const SssIO b = reload_args(a0);   // the output planes' pointers, for the stores
            stg(b.r, i, r);
            st3(b.origin, i, off);
            st3(b.dir, i, dir);
            stg(b.maxdist, i, maxdist);
            stg(b.pdf, i, pdf);
            strgb(b.profile, i, R, G, B);
        } else if (OP == OP_MIS) {
            Frame fr = sss_frame(ld3(a.c.N, i), ld3(a.c.T, i), a.c.has_dPdu != 0);
            stg(a.pdf, i, sss_mis_pdf(p, fr, ld3(a.disp, i), ld3(a.sampleN, i), a.literal != 0));
        }
    }
}

template <int OP, int MODE, int FAST_MATH = RLS_FAST>
__global__ RLS_SSS_ATTR(MODE) void sss_kernel(SssIO a0)
{
    sss_body<OP, MODE, FAST_MATH>(a0);
}

#if RLS_DIAGNOSTICS
template <int OP, int MODE, int FAST_MATH = RLS_FAST>
__global__ RLS_SSS_ATTR(MODE) void sss_kernel_stamped(SssIO a0, unsigned long long *stamps)
{
    ClockStamp<1> cs;
    cs.begin();
    sss_body<OP, MODE, FAST_MATH>(a0);
    cs.end(stamps);
}
#endif


template <int OP, int FAST_MATH = RLS_FAST>
__global__ __launch_bounds__(rlsh::kBlock) void misc_kernel(MiscIO a)
{
    const TileRange tiles = tile_range(a.n);
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const Idx i = make_idx(base);
        if (i.full() >= a.n) continue;
        if (OP == OP_CAVITY) {
            V3 disp = ld3(a.a, i);
            stg(a.out, i, sss_cavity_fade(disp, length(disp), ld3(a.b, i), ld3(a.c, i)));
        } else if (OP == OP_DIFFUSE_DIR) {
            Frame fr;
            fr.N = ld3(a.a, i);
            fr.U = ld3(a.b, i);
            fr.V = cross(fr.N, fr.U);
            st3(a.v0, i, cosine_hemisphere(fr, ldg(a.rx, i), ldg(a.ry, i)));
        } else if (OP == OP_REFLECT_LUM) {
            st3(a.v0, i, reflect_direction(ld3(a.a, i), ld3(a.b, i)));
            V3 c = ld3(a.c, i);
            stg(a.out, i, luminance(c.x, c.y, c.z));
        } else {
            float u = ldg(a.rx, i), v = ldg(a.ry, i);
            st3(a.v0, i, spherical_direction(2.0f * u - 1.0f, kTwoPi * v));
            V2 d = concentric_disk(u, v);
            st3(a.v1, i, mk(d.x, d.y, 0.0f));
        }
    }
}

rls_status check_closure(const rls_sss_closure *c, bool need_frame)
{
    RLS_REQUIRE(c != nullptr, "closure is NULL");
    RLS_REQUIRE(rlsh::ok_rgb(c->sss_color), "sss_color planes must be all set or all NULL");
    RLS_REQUIRE(rlsh::ok_materials(c->materials), "materials.id is set but materials.count is 0");
    if (need_frame) RLS_REQUIRE(rlsh::has3(c->N) && rlsh::has3(c->T), "N/T plane is NULL");
    return RLS_OK;
}

template <int OP>
rls_status launch_kernel(rls_context *ctx, const SssIO &io, const char *name)
{
    const rls_sss_closure &c = io.c;
    const bool uniform = !c.materials.id && !c.sss_dist_multiplier.v && !c.sss_scatter_dist[0].v && !c.sss_scatter_dist[1].v &&
                         !c.sss_scatter_dist[2].v;
    // evalProfile alone uses nothing setDistance computes but maxR: no uniform specialisation of it
    constexpr bool kHoists = OP != OP_ND_EVAL;
#if RLS_DIAGNOSTICS
    if constexpr (OP == OP_PROBE) {      // BASELINE config 4 under rls_diag_clock_stamps_begin: the stamped instantiation
        if (unsigned long long *stamps = (!c.materials.id && !(uniform && kHoists)) ? rlsh::stamps_for_launch(ctx) : nullptr) {
            hipLaunchKernelGGL((sss_kernel_stamped<OP, PER_POINT>), rlsh::grid_for(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io, stamps);
            return rlsh::check_launch(name);
        }
    }
#endif
    if (c.materials.id)
        hipLaunchKernelGGL((sss_kernel<OP, BY_REFERENCE>), rlsh::grid_for(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    else if (uniform && kHoists)
        hipLaunchKernelGGL((sss_kernel<OP, kHoists ? UNIFORM_DISTANCE : PER_POINT>), rlsh::grid_for_hoisting(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    else
        hipLaunchKernelGGL((sss_kernel<OP, PER_POINT>), rlsh::grid_for(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    return rlsh::check_launch(name);
}

template <int OP>
rls_status launch_misc_kernel(rls_context *ctx, const MiscIO &io, const char *name)
{
    hipLaunchKernelGGL(misc_kernel<OP>, rlsh::grid_for(ctx, io.n), dim3(rlsh::kBlock), 0, ctx->stream, io);
    return rlsh::check_launch(name);
}

} // namespace

#if RLS_FAST
RLS_HIDDEN rls_status rls_fast_sss(rls_context *ctx, int op, const rlsh::SssIO *io)
{
    switch (op) {
    case OP_ND: return launch_kernel<OP_ND>(ctx, *io, "rls_nd_sample[fast]");
    case OP_ND_PDF: return launch_kernel<OP_ND_PDF>(ctx, *io, "rls_nd_pdf[fast]");
    case OP_ND_EVAL: return launch_kernel<OP_ND_EVAL>(ctx, *io, "rls_nd_eval[fast]");
    case OP_PROBE: return launch_kernel<OP_PROBE>(ctx, *io, "rls_sss_probe_ray[fast]");
    default: return launch_kernel<OP_MIS>(ctx, *io, "rls_sss_mis_pdf[fast]");
    }
}
RLS_HIDDEN rls_status rls_fast_misc(rls_context *ctx, int op, const rlsh::MiscIO *io)
{
    switch (op) {
    case OP_CAVITY: return launch_misc_kernel<OP_CAVITY>(ctx, *io, "rls_sss_cavity_fade[fast]");
    case OP_DIFFUSE_DIR: return launch_misc_kernel<OP_DIFFUSE_DIR>(ctx, *io, "rls_sss_sample_diffuse_direction[fast]");
    case OP_REFLECT_LUM: return launch_misc_kernel<OP_REFLECT_LUM>(ctx, *io, "rls_util_reflect_luminance[fast]");
    default: return launch_misc_kernel<OP_UTIL>(ctx, *io, "rls_util_directions[fast]");
    }
}
#else
RLS_HIDDEN rls_status rls_fast_sss(rls_context *ctx, int op, const rlsh::SssIO *io);
RLS_HIDDEN rls_status rls_fast_misc(rls_context *ctx, int op, const rlsh::MiscIO *io);

namespace {
template <int OP>
rls_status launch(rls_context *ctx, const SssIO &io, const char *name)
{
    return ctx->fast ? rls_fast_sss(ctx, OP, &io) : launch_kernel<OP>(ctx, io, name);
}
template <int OP>
rls_status launch_misc(rls_context *ctx, const MiscIO &io, const char *name)
{
    return ctx->fast ? rls_fast_misc(ctx, OP, &io) : launch_misc_kernel<OP>(ctx, io, name);
}
} // namespace

#define RLS_PROLOGUE(frame)                              \
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");          \
    RLS_REQUIRE(n >= 0, "n < 0");                        \
    if (n == 0) return RLS_OK;                           \
    { rls_status _s = check_closure(c, frame); if (_s != RLS_OK) return _s; }

extern "C" {

rls_status rls_nd_sample(rls_context *ctx, int64_t n, const rls_sss_closure *c, const float *rx,
                         float *r, float *pdf, rls_rgb profile)
{
    RLS_PROLOGUE(false);
    RLS_REQUIRE(rx && r && pdf && rlsh::has3(profile), "NULL plane");
    SssIO io = {};
    io.c = *c; io.rx = rx; io.r = r; io.pdf = pdf; io.profile = profile; io.n = n;
    return launch<OP_ND>(ctx, io, "rls_nd_sample");
}

rls_status rls_nd_pdf(rls_context *ctx, int64_t n, const rls_sss_closure *c, const float *r, float *pdf)
{
    RLS_PROLOGUE(false);
    RLS_REQUIRE(r && pdf, "NULL plane");
    SssIO io = {};
    io.c = *c; io.rin = r; io.pdf = pdf; io.n = n;
    return launch<OP_ND_PDF>(ctx, io, "rls_nd_pdf");
}

rls_status rls_nd_eval(rls_context *ctx, int64_t n, const rls_sss_closure *c, const float *r, rls_rgb profile)
{
    RLS_PROLOGUE(false);
    RLS_REQUIRE(r && rlsh::has3(profile), "NULL plane");
    SssIO io = {};
    io.c = *c; io.rin = r; io.profile = profile; io.n = n;
    return launch<OP_ND_EVAL>(ctx, io, "rls_nd_eval");
}

rls_status rls_sss_probe_ray(rls_context *ctx, int64_t n, const rls_sss_closure *c,
                             const float *rx, const float *ry, rls_cvec3 P,
                             float *r, rls_vec3 origin, rls_vec3 dir, float *maxdist,
                             float *pdf, rls_rgb profile)
{
    RLS_PROLOGUE(true);
    RLS_REQUIRE(rx && ry, "rx/ry is NULL");
    RLS_REQUIRE(rlsh::has3(P) || rlsh::none3(P), "P planes must be all set or all NULL");
    RLS_REQUIRE(r && rlsh::has3(origin) && rlsh::has3(dir) && maxdist && pdf && rlsh::has3(profile), "NULL output plane");
    SssIO io = {};
    io.c = *c; io.rx = rx; io.ry = ry; io.P = P; io.r = r; io.origin = origin; io.dir = dir;
    io.maxdist = maxdist; io.pdf = pdf; io.profile = profile; io.n = n;
    return launch<OP_PROBE>(ctx, io, "rls_sss_probe_ray");
}

rls_status rls_sss_mis_pdf(rls_context *ctx, int64_t n, const rls_sss_closure *c,
                           rls_cvec3 disp, rls_cvec3 sampleN, int literal_matrix, float *pdf)
{
    RLS_PROLOGUE(true);
    RLS_REQUIRE(rlsh::has3(disp) && rlsh::has3(sampleN) && pdf, "NULL plane");
    SssIO io = {};
    io.c = *c; io.disp = disp; io.sampleN = sampleN; io.literal = literal_matrix; io.pdf = pdf; io.n = n;
    return launch<OP_MIS>(ctx, io, "rls_sss_mis_pdf");
}

rls_status rls_sss_cavity_fade(rls_context *ctx, int64_t n, rls_cvec3 disp, rls_cvec3 sampleN,
                               rls_cvec3 No, float *fade)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(rlsh::has3(disp) && rlsh::has3(sampleN) && rlsh::has3(No) && fade, "NULL plane");
    MiscIO io = {};
    io.a = disp; io.b = sampleN; io.c = No; io.out = fade; io.n = n;
    return launch_misc<OP_CAVITY>(ctx, io, "rls_sss_cavity_fade");
}

rls_status rls_sss_sample_diffuse_direction(rls_context *ctx, int64_t n, rls_cvec3 normal, rls_cvec3 T,
                                            const float *rx, const float *ry, rls_vec3 wi)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(rlsh::has3(normal) && rlsh::has3(T) && rx && ry && rlsh::has3(wi), "NULL plane");
    MiscIO io = {};
    io.a = normal; io.b = T; io.rx = rx; io.ry = ry; io.v0 = wi; io.n = n;
    return launch_misc<OP_DIFFUSE_DIR>(ctx, io, "rls_sss_sample_diffuse_direction");
}

rls_status rls_util_directions(rls_context *ctx, int64_t n, const float *a, const float *b,
                               rls_vec3 spherical, rls_vec3 disk)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(a && b && rlsh::has3(spherical) && rlsh::has3(disk), "NULL plane");
    MiscIO io = {};
    io.rx = a; io.ry = b; io.v0 = spherical; io.v1 = disk; io.n = n;
    return launch_misc<OP_UTIL>(ctx, io, "rls_util_directions");
}

rls_status rls_util_reflect_luminance(rls_context *ctx, int64_t n, rls_cvec3 i, rls_cvec3 nrm, rls_cvec3 color,
                                      rls_vec3 reflected, float *luminance)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(rlsh::has3(i) && rlsh::has3(nrm) && rlsh::has3(color) && rlsh::has3(reflected) && luminance, "NULL plane");
    MiscIO io = {};
    io.a = i; io.b = nrm; io.c = color; io.v0 = reflected; io.out = luminance; io.n = n;
    return launch_misc<OP_REFLECT_LUM>(ctx, io, "rls_util_reflect_luminance");
}

} // extern "C"

#endif // !RLS_FAST
