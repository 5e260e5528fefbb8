/*
 * rlshaders_amd.h -- C ABI of the MI355X-native batched BSDF evaluator / importance sampler.
 *
 * This is the drop-in boundary for the closure layer of shihchinw/rlShaders: the per-shading-point
 * (sample, eval, pdf) callback triple that the reference hands to Arnold
 *     static AtVector evalSample(const void *brdf, float rx, float ry);
 *     static AtColor  evalBrdf  (const void *brdf, const AtVector *indir);
 *     static float    evalPdf   (const void *brdf, const AtVector *indir);
 * (src/rlGgx.h:97,110,121; src/rlDisney.cpp:109,120,139 -- paths relative to the reference
 * repository), turned into batch calls over planar SoA device arrays, one entry per shading point.
 * The closure object the reference builds on the stack per point (`brdf`) becomes a struct of
 * device pointers named after the reference's node parameters (src/rlShaders.mtd,
 * node_parameters in src/rlGgx.cpp:170-198, src/rlDisney.cpp:604-638, src/rlSkin.cpp:107-139);
 * construction (src/rlGgx.h:130-156, src/rlDisney.cpp:155-192, src/rlSss.cpp:20-34) happens
 * in registers inside the kernels.
 *
 * Conventions (all kept from the reference):
 *   - eval returns BRDF x signed cosine (src/rlGgx.h:164, src/rlDisney.cpp:133,136);
 *   - an invalid sample is the zero vector; eval of a zero vector is black, pdf of it is 0
 *     (src/rlDisney.cpp:124-127,141-144,385-387; src/rlGgx.h:112-115);
 *   - the GGX pdf is floored at 1e-4 (src/rlGgx.h:79), the Disney diffuse pdf too
 *     (src/rlDisney.cpp:517);
 *   - no exceptions cross this ABI; every entry point returns an rls_status.
 *
 * Geometry inputs per point: wo = -sg->Rd, N = sg->Nf (face-forward), T = the tangent U that
 * AiBuildLocalFramePolar(&U,&V,&N) returns (closed Arnold code, so an input; V = N x T).
 * All pointers are DEVICE pointers owned by the caller; a plane holds n floats; planes of a
 * vec3/rgb need not be adjacent.  The library allocates nothing per call.
 * All arithmetic is fp32.  Launches go to the context's stream and are asynchronous.
 *
 * Threading: a context is a (device, stream, math mode) triple with no internal locking -- use one
 * context per host thread (any number per device; rlshaders_amd/host/example_multi_gpu.cpp runs eight
 * on one GPU concurrently).  Closure structs, arenas and graphs are plain data / handles and may be
 * shared between threads as long as the device memory they name is not written concurrently.
 * rls_last_error() is per thread.
 */
#ifndef RLSHADERS_AMD_H
#define RLSHADERS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLS_VERSION_MAJOR 0
#define RLS_VERSION_MINOR 1

typedef int rls_status;
enum {
    RLS_OK = 0,
    RLS_ERR_INVALID_ARGUMENT = 1,   /* NULL required pointer, n < 0, bad enum          */
    RLS_ERR_NO_DEVICE = 2,          /* no HIP device / bad ordinal                      */
    RLS_ERR_HIP = 3,                /* a HIP runtime call failed; see rls_last_error()  */
    RLS_ERR_OUT_OF_MEMORY = 4,
    RLS_ERR_UNSUPPORTED = 5,
    RLS_ERR_ABORTED = 6             /* a caller-supplied callback asked to stop (rls_disney_integrate_chunked) */
};

typedef struct rls_context rls_context;

/* planar vec3 / rgb views (device pointers) */
typedef struct { const float *x, *y, *z; } rls_cvec3;
typedef struct { float *x, *y, *z; } rls_vec3;
typedef struct { const float *r, *g, *b; } rls_crgb;
typedef struct { float *r, *g, *b; } rls_rgb;

/* A node parameter: per-point stream (v != NULL, n floats) or one uniform value (v == NULL).
 * Arnold parameters are constants unless a texture is linked; uniform ones cost no bandwidth, and a closure whose
 * parameters are ALL uniform runs kernels that evaluate the parameter-only arithmetic once per thread instead of once
 * per point (6 - 20 % less time for the one-sample verbs; the results are the same bits either way). */
typedef struct { const float *v; float u; } rls_param;
typedef struct { const float *r, *g, *b; float ur, ug, ub; } rls_param_rgb;

/* Node parameters BY REFERENCE (a "material table"): a batch that mixes the hits of many node instances need not carry
 * the parameters per point.  With `id != NULL` in a closure's `materials` member every non-NULL parameter pointer of that
 * closure (and of the rls_ggx_shader handed over with it) is a COLUMN of `count` floats -- one entry per node instance --
 * and point i takes entry min(id[i], count - 1): 4 bytes per point instead of 4 per parameter, which is what a host-resident
 * batch pays for on the bus (rls_pipeline_*).  A NULL parameter pointer is still the uniform value.  The arithmetic is
 * the per-point arithmetic on the looked-up values: the same bits as with the values expanded into planes.
 * `id == NULL` (a zero-initialised closure): parameters are per-point planes or uniform values, as before.
 * No counterpart in the reference, which evaluates the parameters per hit (AiShaderEvalParam* at src/rlGgx.cpp:256-259,
 * src/rlDisney.cpp:155-172, src/rlSkin.cpp:184-236): this is how a batching stub avoids expanding them per point. */
typedef struct { const uint32_t *id; uint32_t count; } rls_material_index;

/* Arnold ray-type tags selecting the rlDisney lobe (DisneySampler::mSampleType,
 * src/rlDisney.cpp:112,132,147,194-197) */
#define RLS_RAY_DIFFUSE 0x08
#define RLS_RAY_GLOSSY  0x10

/* microfacet-normal sampling kernels of rlGgx (src/rlGgx.h:24-89; the reference selects VNDF,
 * src/rlGgx.h:375) */
#define RLS_KERNEL_VNDF 0
#define RLS_KERNEL_NDF  1

/* ------------------------------------------------------------------------------------------
 * Context, memory, timing
 * ---------------------------------------------------------------------------------------- */
/* Visible HIP devices (0 when there is none or the runtime cannot initialise). */
int         rls_device_count(void);
/* The contiguous shard [first, first + count) of rank `rank` of `world` over `total` shading points; the shards
 * tile [0, total) exactly.  Points are independent, so a batch shards by index range with no collective on the
 * data path (one context per device). */
rls_status  rls_shard_range(int64_t total, int rank, int world, int64_t *first, int64_t *count);
rls_status  rls_context_create(int device_ordinal, rls_context **out);
void        rls_context_destroy(rls_context *ctx);
/* Launch on an existing hipStream_t (e.g. the framework's current stream).  The handle is taken
 * literally: NULL is HIP's default (null) stream.  A new context launches on a private
 * non-blocking stream until this is called; rls_context_use_own_stream() goes back to it. */
rls_status  rls_context_set_stream(rls_context *ctx, void *hip_stream);
rls_status  rls_context_use_own_stream(rls_context *ctx);
void       *rls_context_get_stream(rls_context *ctx);
/* Arithmetic of the closure kernels launched through this context.
 *   RLS_MATH_EXACT (default): IEEE division, correctly rounded sqrt, and elementary functions
 *     (atan2f acosf tanf sinf cosf expf logf powf) that restate the host libm's (glibc) algorithms --
 *     closure outputs agree with the CPU closures bit for bit.
 *   RLS_MATH_FAST: hardware reciprocal / sqrt / sin / cos / exp grade arithmetic (~1 ulp per
 *     operation) and the visible-normal view analysis by vector algebra instead of the reference's
 *     atan2f/acosf/tanf round trip.  Same formulas, same conventions; outputs within 1e-5 of the CPU
 *     closures wherever those are well conditioned.  Roughly 3x fewer
 *     instructions: the kernels become HBM-bandwidth-bound. */
#define RLS_MATH_EXACT 0
#define RLS_MATH_FAST  1
rls_status  rls_context_set_math_mode(rls_context *ctx, int mode);
int         rls_context_get_math_mode(const rls_context *ctx);
rls_status  rls_context_synchronize(rls_context *ctx);
int         rls_context_device(const rls_context *ctx);
/* Thread-local text of the most recent failure in the calling thread ("" if none). */
const char *rls_last_error(void);
const char *rls_status_string(rls_status s);
int         rls_version(void);                 /* major*1000 + minor */
/* Which host libm the EXACT kernels reproduce bit for bit.  The reference's closures call sinf / cosf / expf / logf / powf /
 * atan2f / acosf / tanf of the C library the renderer runs on (src/rlGgx.cpp:27-58, src/rlDisney.cpp:177,399,549,576,
 * src/rlSss.cpp:31-32,59,62,78-79,102); this library restates glibc's algorithms (verified range 2.28 <= glibc < 2.41: the
 * code was read from the disassembly of 2.35; glibc 2.41 replaced tanf / acosf / atan2f and others by correctly rounded
 * CORE-MATH versions, and a host with such a glibc reports mismatches > 0 through the tanf / acosf / atan2f probes below;
 * provenance and licences: THIRD_PARTY.md), and of glibc's two x86-64 builds the one chosen when the library was compiled:
 * rls_libm_flavour() returns "glibc-fma" (the build glibc's ifunc selects on every CPU with AVX2 + FMA) or "glibc-sse2".
 * rls_host_libm_matches() asks the CALLER's libm (the one this process resolves): the 42 fp32 arguments on which the two
 * glibc builds differ -- every such argument of sinf / cosf / expf / powf(x, 5), found by sweeping all 2^32 -- must give the
 * followed build's result, and this library's routines compiled for the host must agree with the host's on 4096 arguments
 * per function (a libm that is neither build).  *mismatches = the number of disagreements.  "Bit for bit like the CPU
 * closures" holds on THIS host if and only if it is 0.  Otherwise the elementary functions differ in the last bit on a
 * ~1e-8 ... 1e-6 share of arguments (glibc's other build) or more (a newer glibc, musl, MSVC's UCRT -- the reference
 * author's platform, rlShaders.sln), and chained closure outputs show SURVEY.md Appendix D's alternate-libm tail: 0.009-0.14 %
 * of them beyond 1e-5 relative -- the reference disagreeing with itself across platforms.  Needs no device.
 * (The library builds in either flavour -- `python -m rlshaders_amd.build --variant sse2 -DRLM_GLIBC_FMA=0` -- and the SSE2
 * flavour is checked against glibc's SSE2 build the same way: 0 differences on all 2^32 arguments per function.) */
const char *rls_libm_flavour(void);
rls_status  rls_host_libm_matches(int *mismatches);
/* Device properties the host side sizes shards with. */
rls_status  rls_device_info(rls_context *ctx, int *compute_units, size_t *hbm_bytes_total,
                            size_t *hbm_bytes_free, char *arch_name, size_t arch_name_len);

rls_status  rls_device_alloc(rls_context *ctx, size_t bytes, void **out);
rls_status  rls_device_free(rls_context *ctx, void *p);
rls_status  rls_copy_to_device(rls_context *ctx, void *dst, const void *src_host, size_t bytes);
rls_status  rls_copy_to_host(rls_context *ctx, void *dst_host, const void *src, size_t bytes);

/* Plane arenas.  Where in HBM the planes of a batch live matters: the closure kernels stream ~31 planes at once,
 * and (a) planes carved from ONE allocation run the reflect+refract kernel 1-2 % faster than 31 separate
 * allocations do, (b) equally sized blocks differ by up to 18 % in the bandwidth that plane
 * pattern reaches on them, stably for the life of the allocation ("Placement").  An arena is one
 * device allocation carved into `planes` planes of n floats; with candidates > 1 that many blocks are allocated,
 * each timed with an arithmetic-free copy of the kernels' access pattern, and the fastest is kept.
 * rls_probe_block times a caller-owned block the same way (its contents are overwritten).  Both synchronise. */
typedef struct rls_arena rls_arena;
rls_status  rls_arena_create(rls_context *ctx, int64_t n, int planes, int candidates, rls_arena **out);
float      *rls_arena_plane(const rls_arena *arena, int k);                    /* NULL if k is out of range */
rls_status  rls_arena_info(const rls_arena *arena, size_t *bytes, int *candidates_probed, float *probe_gb_per_s,
                           float *probe_min, float *probe_max);
void        rls_arena_destroy(rls_arena *arena);
rls_status  rls_probe_block(rls_context *ctx, void *block, size_t bytes, float *gb_per_s);

/* HIP-event stopwatch on the context's stream (what bench.py times kernels with). */
rls_status  rls_timer_start(rls_context *ctx);
rls_status  rls_timer_stop(rls_context *ctx);
rls_status  rls_timer_elapsed_ms(rls_context *ctx, float *ms);   /* synchronises on the stop event */

/* Measurement aids (the in-kernel clock stamps bench.py reads the shader clock with) are NOT part of the drop-in surface:
 * they are declared in rlshaders_amd_diag.h; a plugin never needs them. */

/* Launch graphs.  A renderer that flushes small batches (a bucket's worth of shading points) through
 * the same device buffers again and again is launch-bound, not kernel-bound: record the rls_* calls of
 * one flush once and replay them as a single HIP graph launch.  Between begin and end every closure /
 * integrator / generator entry point called on this context from this thread is recorded instead
 * of executed (same arguments, same math mode); calls that synchronise with the host --
 * rls_context_synchronize, rls_copy_*, rls_checksum, rls_timer_elapsed_ms, rls_device_alloc/free --
 * are not allowed while recording.  The context must be on a real stream (its own, or one given to
 * rls_context_set_stream), not on the NULL stream.  No reference counterpart (Arnold calls the
 * closures inline, src/rlGgx.cpp:286-295). */
typedef struct rls_graph rls_graph;
rls_status  rls_graph_begin_capture(rls_context *ctx);
rls_status  rls_graph_end_capture(rls_context *ctx, rls_graph **out);
rls_status  rls_graph_launch(rls_context *ctx, rls_graph *graph);      /* asynchronous, on the context's stream */
void        rls_graph_destroy(rls_graph *graph);

/* Host-resident batches.  The reference's closures run per hit on CPU render threads (src/rlGgx.cpp:248-261 builds the
 * closure on shader_evaluate's stack), so the shading points an Arnold-side stub gathers start in HOST memory and the
 * results are wanted there.  rls_host_alloc gives page-locked host memory (DMA at the full PCIe rate, asynchronous
 * copies); rls_host_register page-locks memory the host already owns (its batch buffers, once).
 * A pipeline cuts a batch of n points into chunks of chunk_points and sends them through `depth` slots, each slot a
 * stream of its own with its own device planes (in_planes + out_planes planes of chunk_points floats): per chunk the
 * input planes are uploaded from host_in[k] + first_point, `launch` is called with the SLOT's context and device planes
 * -- it enqueues the closure calls for `count` points on that context, e.g. rls_ggx_reflect_refract(slot, count, ...) --
 * and the output planes are downloaded to host_out[k] + first_point.  With depth >= 2 the upload of chunk k + 1, the
 * kernels of chunk k and the download of chunk k - 1 overlap.  A NULL host plane is skipped (a uniform parameter, an
 * unwanted output).  Host planes that are equally spaced inside ONE allocation (a [planes, n] array: rlsb::HostPlanes)
 * travel as one strided copy per chunk and direction -- a third faster than one copy per plane.  The host planes SHOULD be
 * page-locked (rls_host_alloc / rls_host_register): pageable memory is accepted, but the runtime then stages every copy
 * through its own pinned buffer and blocks the calling thread meanwhile -- the results are the same, the overlap is gone
 * (the Python wrapper refuses pageable tensors for that reason).  rls_pipeline_run returns when every chunk has arrived in host memory; the slots compute in the
 * arithmetic mode of the context the pipeline was created on.  Results are those of the device-resident call on the
 * same points, bit for bit.  PCIe-bound by construction: rls_measure_copy_rates gives the box's pinned-memory rates
 * (host -> device, device -> host, both at once; GB/s) to hold a pipeline's throughput against. */
rls_status  rls_host_alloc(rls_context *ctx, size_t bytes, void **out);
rls_status  rls_host_free(rls_context *ctx, void *p);
rls_status  rls_host_register(rls_context *ctx, void *p, size_t bytes);
rls_status  rls_host_unregister(rls_context *ctx, void *p);
typedef struct rls_pipeline rls_pipeline;
/* returns an rls_status (RLS_OK to go on) */
typedef int (*rls_pipeline_launch_fn)(void *user, rls_context *slot, int64_t first_point, int64_t count,
                                      float *const *device_in, float *const *device_out);
rls_status  rls_pipeline_create(rls_context *ctx, int64_t chunk_points, int in_planes, int out_planes, int depth,
                                rls_pipeline **out);
rls_status  rls_pipeline_run(rls_pipeline *p, int64_t n, const float *const *host_in, float *const *host_out,
                             rls_pipeline_launch_fn launch, void *user);
void        rls_pipeline_destroy(rls_pipeline *p);
rls_status  rls_measure_copy_rates(rls_context *ctx, size_t bytes, float rates_gb_per_s[3]);

/* ------------------------------------------------------------------------------------------
 * rlGgx closure: rls::GgxSamplerT<VNDFKernel>  (src/rlGgx.h:92-373, src/rlGgx.cpp:14-99)
 * Parameter names: src/rlGgx.cpp:172-186 (KsColor, specularRoughness, ior, anisotropic).
 * ---------------------------------------------------------------------------------------- */
typedef struct rls_ggx_closure {
    rls_cvec3      wo, N, T;
    const uint8_t *exiting;            /* optional; 1 where !(dot(sg->N, sg->Rd) < 1e-4): swaps
                                          the IORs as src/rlGgx.h:137-142 does                 */
    rls_param_rgb  KsColor;            /* specColor                                            */
    rls_param      specularRoughness;  /* roughness; alpha = roughness^2 (src/rlGgx.h:149)     */
    rls_param      ior;
    rls_param      anisotropic;
    rls_material_index materials;      /* optional: the parameters above as per-material columns */
} rls_ggx_closure;

/* evalSample (src/rlGgx.h:97-107): wi = reflect(wo, VNDF microfacet).  fresnel (optional)
 * receives the Fresnel term the reference accumulates into mReflectWeight for this sample. */
rls_status rls_ggx_sample(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                          const float *rx, const float *ry, rls_vec3 wi, float *fresnel);
/* evalBrdf (src/rlGgx.h:110-119,158-165,304-313) */
rls_status rls_ggx_eval(rls_context *ctx, int64_t n, const rls_ggx_closure *c, rls_cvec3 wi, rls_rgb f);
/* evalPdf (src/rlGgx.h:121-127,72-80) */
rls_status rls_ggx_pdf(rls_context *ctx, int64_t n, const rls_ggx_closure *c, rls_cvec3 wi, float *pdf);
/* the triple fused in the reference's call order: sample -> eval(wi) -> pdf(wi) */
rls_status rls_ggx_sample_eval_pdf(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                                   const float *rx, const float *ry,
                                   rls_vec3 wi, rls_rgb f, float *pdf, float *fresnel);
/* per-sample body of integrateRefract (src/rlGgx.h:228-242): microfacet sample, Snell refraction
 * about it (mirror on total internal reflection), weight = getSampleWeight (src/rlGgx.h:294-301).
 * refracted (optional) gets 1 / 0 (TIR). */
rls_status rls_ggx_refract_sample(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                                  const float *rx, const float *ry,
                                  rls_vec3 wt, float *weight, uint8_t *refracted);
/* reflect triple with (rx,ry) + refract sample with (rx2,ry2) over the same closure, one pass */
rls_status rls_ggx_reflect_refract(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                                   const float *rx, const float *ry, const float *rx2, const float *ry2,
                                   rls_vec3 wi, rls_rgb f, float *pdf, float *fresnel,
                                   rls_vec3 wt, float *weight);
/* microfacet normal only: kernel = RLS_KERNEL_VNDF (src/rlGgx.cpp:63-99) or RLS_KERNEL_NDF
 * (src/rlGgx.h:33-41) */
rls_status rls_ggx_microfacet(rls_context *ctx, int64_t n, const rls_ggx_closure *c, int kernel,
                              const float *rx, const float *ry, rls_vec3 m);
/* NDFKernel::evalPdf (src/rlGgx.h:45-50), the alternate pdf */
rls_status rls_ggx_ndf_pdf(rls_context *ctx, int64_t n, const rls_ggx_closure *c, rls_cvec3 wi, float *pdf);
/* spp_n^2 stratified samples per point drawn in-kernel (stand-in for AiSampler(spp_n, 2),
 * src/rlGgx.cpp:148): sum_f_over_pdf = sum over samples of eval/pdf (what AiBRDFIntegrate
 * accumulates before radiance), avg_reflect_weight = getAvgReflectWeight (src/rlGgx.h:181-184).
 * Small batches are given 4, 16 or 64 lanes per point (chosen from n, so that they still fill the GPU); the sums grow in
 * SAMPLE order whatever that width (the lanes' terms are folded into a replicated sum in lane order, round by round), so
 * the results do not depend on it: bit for bit those of one lane per point, i.e. of the reference's `result +=` loop.
 * first_index (here and in every other in-kernel-sampling entry point): the global index of point 0 of
 * this call.  The per-point scrambles are hash(seed, first_index + i), so a batch split with
 * rls_shard_range -- or walked in chunks -- draws exactly the numbers of the unsplit batch and, by the above, returns
 * exactly its sums. */
rls_status rls_ggx_integrate(rls_context *ctx, int64_t n, const rls_ggx_closure *c,
                             int spp_n, uint32_t seed, uint64_t first_index,
                             rls_rgb sum_f_over_pdf, float *avg_reflect_weight);

/* integrateRefract (src/rlGgx.h:205-245).  traced != 0: the sample loop of 228-244 -- per sample a microfacet
 * normal, the refraction of the view about it (Snell, Walter et al. eq. 40; the mirror direction on total
 * internal reflection, 234-237), radiance x getSampleWeight (294-301), the sum x AiSamplerGetSampleInvCount.
 * traced == 0: the branch of 213-222 -- one refraction about the shading normal, radiance x SQR(iorOut / iorIn)
 * x |Nf . dir|, black on total internal reflection.  AiTrace / AiTraceBackground are closed: the radiance is that
 * of a uniform environment, env[3] (parity unpinned).  tir_fraction (optional): the share of samples that were
 * totally internally reflected. */
rls_status rls_ggx_integrate_refract(rls_context *ctx, int64_t n, const rls_ggx_closure *c, int traced,
                                     const float env[3], int spp_n, uint32_t seed, uint64_t first_index,
                                     rls_rgb result, float *tir_fraction);

/* Direct lighting of the rlGgx node: the light loop of shader_evaluate (src/rlGgx.cpp:274-299) --
 *     diffuse  += AiEvaluateLightSample(sg, diffData, AiOrenNayarMISSample, ..BRDF, ..PDF)
 *     specular += sampler.evalLightSample(sg)        (AiEvaluateLightSample over the GGX triple,
 *                                                      src/rlGgx.h:167-170)
 * followed by diffuse *= KdColor * Kd and specular *= Ks: the two direct AOVs (src/rlGgx.cpp:307-308).
 * Arnold's light loop, AiEvaluateLightSample and the Oren-Nayar MIS closure are closed; documented
 * stand-ins (parity unpinned): one spherical area light sampled uniformly over the cone it subtends
 * from sg->P, no occluders; the qualitative Oren-Nayar model (SIGGRAPH'94) with cosine-weighted
 * sampling about N; the two-sample estimator with the power heuristic w_a = p_a^2 / (p_a^2 + p_b^2)
 * over spp_n^2 light samples (shared by both lobes) and spp_n^2 BSDF samples per lobe; directions below
 * the shading normal contribute nothing.  The per-point partial sums of the G lanes that share a
 * shading point are reduced with wave shuffles.  mis_mode selects the estimator: both strategies, light
 * samples only, BSDF samples only -- equal in expectation when sample / eval / pdf are consistent.
 * `lights` is an array of n_lights (1 .. RLS_MAX_LIGHTS) lights: the loop `while (AiLightsGetSample(sg))` visits
 * every sample of every light, so light l runs the estimator above with its own sample streams (3 l .. 3 l + 2) and
 * the AOVs are the sums over the lights, added in array order.  Within a light each strategy (light samples, BSDF
 * samples) keeps its own sum, grown in sample order; the two are added at the end. */
#define RLS_MAX_LIGHTS     8
#define RLS_MIS_BOTH       0
#define RLS_MIS_LIGHT_ONLY 1
#define RLS_MIS_BSDF_ONLY  2
typedef struct rls_sphere_light {
    float center[3], radius;
    float radiance[3];
    int   mis_mode;
} rls_sphere_light;
/* the node parameters of shader_evaluate that are not part of the specular closure (src/rlGgx.cpp:170-179);
 * the light loop reads the first four, rls_ggx_shade all of them */
typedef struct rls_ggx_shader {
    rls_param_rgb KdColor;
    rls_param     Kd, diffuseRoughness, Ks;
    rls_param_rgb KtColor;
    rls_param     Kt;
} rls_ggx_shader;
rls_status rls_ggx_direct_lighting(rls_context *ctx, int64_t n, const rls_ggx_closure *c, const rls_ggx_shader *sh,
                                   rls_cvec3 P, const rls_sphere_light *lights, int n_lights, int spp_n, uint32_t seed,
                                   uint64_t first_index, rls_rgb direct_diffuse, rls_rgb direct_specular);

/* shader_evaluate of the rlGgx node for a camera ray, whole (src/rlGgx.cpp:248-327), in one pass over one closure
 * set-up:
 *     the light loop (285-299) -> diffuse *= KdColor * Kd, specular *= Ks                       (304-305)
 *     transmission = AiColorIsSmall(KtColor * Kt) ? black : integrateRefract * (KtColor * Kt)    (307-309)
 *     indirectDiffuse = sampleDiffuse ? (KdColor * Kd) * AiBRDFIntegrate(Oren-Nayar) : black     (315-319)
 *     indirectGlossy  = integrateGlossy * Ks                                                     (321)
 *     sg->out.RGB = (diffuse + specular + transmission) + (indirectDiffuse + indirectGlossy)     (311, 323)
 * with sampleDiffuse = !AiColorIsSmall(KdColor * Kd) (280; the ray-depth tests are the caller's: this is depth 0).
 * Opacity (250-254, 326) and the shadow-ray branch (264-269) read no closure and stay with the caller.  The closed
 * renderer services are supplied as in the single-purpose entry points, each loop with spp_n^2 samples (parity
 * unpinned): the light loop as rls_ggx_direct_lighting (n_lights may be 0), integrateRefract as
 * rls_ggx_integrate_refract (`traced`, `env`), AiBRDFIntegrate as the mean of brdf / pdf over the samples times the
 * radiance `env` of a uniform environment (cosine-weighted samples for the Oren-Nayar closure, the GGX triple for
 * integrateGlossy, which returns black for a small KsColor, src/rlGgx.h:174-176).  Sample streams: light l uses
 * 3 l .. 3 l + 2, integrateGlossy 24, integrateRefract 25, the indirect diffuse loop 26. */
typedef struct rls_ggx_shade_out {
    rls_rgb direct_diffuse, direct_specular, refraction, indirect_diffuse, indirect_specular;   /* the AOVs, 314-316, 324-325 */
    rls_rgb out;                                                                                /* optional: sg->out.RGB     */
} rls_ggx_shade_out;
rls_status rls_ggx_shade(rls_context *ctx, int64_t n, const rls_ggx_closure *c, const rls_ggx_shader *sh, rls_cvec3 P,
                         const rls_sphere_light *lights /* NULL when n_lights == 0 */, int n_lights,
                         const float env[3], int traced, int spp_n, uint32_t seed, uint64_t first_index,
                         const rls_ggx_shade_out *out);

/* ------------------------------------------------------------------------------------------
 * rlDisney closure: DisneySampler (src/rlDisney.cpp:105-602)
 * Parameter names: src/rlDisney.cpp:606-610.
 * ---------------------------------------------------------------------------------------- */
typedef struct rls_disney_closure {
    rls_cvec3     wo, N, T;
    rls_param_rgb base_color;
    rls_param     subsurface, metallic, specular, specular_tint, roughness, anisotropic,
                  sheen, sheen_tint, clearcoat, clearcoat_gloss;
    rls_material_index materials;       /* optional: the parameters above as per-material columns */
} rls_disney_closure;

/* lobe = RLS_RAY_DIFFUSE or RLS_RAY_GLOSSY (what setSampleType selects) */
rls_status rls_disney_sample(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                             const float *rx, const float *ry, rls_vec3 wi);
rls_status rls_disney_eval(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                           rls_cvec3 wi, rls_rgb f);
rls_status rls_disney_pdf(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                          rls_cvec3 wi, float *pdf);
rls_status rls_disney_sample_eval_pdf(rls_context *ctx, int64_t n, const rls_disney_closure *c, int lobe,
                                      const float *rx, const float *ry,
                                      rls_vec3 wi, rls_rgb f, float *pdf);
/* spp_n^2 stratified samples per point and lobe, drawn in-kernel, both lobes.
 * Reduced mode: per point and lobe the sum of eval/pdf over valid samples and the valid count
 * (pdf > 1e-4 and wi != 0, as src/rlDisney.cpp:309).  Streamed mode (any of s_* non-NULL):
 * additionally every sample's (wi, f, pdf) at index  lobe*n*spp + s*n + i  (sample-major planes). */
typedef struct rls_disney_stream_out { rls_vec3 wi; rls_rgb f; float *pdf; } rls_disney_stream_out;
rls_status rls_disney_integrate(rls_context *ctx, int64_t n, const rls_disney_closure *c,
                                int spp_n, uint32_t seed, uint64_t first_index,
                                rls_rgb diffuse_sum, float *diffuse_count,
                                rls_rgb specular_sum, float *specular_count,
                                const rls_disney_stream_out *stream /* optional */);
/* Streamed mode over a batch whose samples do not fit in memory at once (BASELINE config 3: 2^26 points x
 * 2 x 64 triples x 28 B = 241 GB).  The point range is walked in chunks of chunk_points; every chunk writes
 * its samples to the SAME chunk buffers (index lobe*count*spp + s*count + k, k = point - first point of the
 * chunk, count = points in this chunk) and the reduced sums to their place in the whole-batch planes.
 * `consume` plays the part of the loop body that uses each sample in the reference (the AiTrace of the
 * explicit sample loop, src/rlDisney.cpp:299-312): it is called on the host right after a chunk's kernel has
 * been enqueued; work it enqueues on the context's stream runs after the chunk is complete and before the
 * next chunk overwrites the buffers.  NULL discards the samples (measurement).  A non-zero return stops the
 * walk and the call returns RLS_ERR_ABORTED.  A consumer is a host-side effect per chunk: it cannot be recorded
 * into a launch graph (a replay would overwrite the chunk buffers back to back without calling it), so between
 * rls_graph_begin_capture and rls_graph_end_capture a call with a consumer is refused (RLS_ERR_UNSUPPORTED).
 * The samples are those of the unchunked rls_disney_integrate call. */
typedef int (*rls_disney_chunk_fn)(void *user, int64_t first_point, int64_t count,
                                   const rls_disney_stream_out *chunk);
rls_status rls_disney_integrate_chunked(rls_context *ctx, int64_t n, const rls_disney_closure *c,
                                        int spp_n, uint32_t seed, uint64_t first_index,
                                        rls_rgb diffuse_sum, float *diffuse_count,
                                        rls_rgb specular_sum, float *specular_count,
                                        int64_t chunk_points, const rls_disney_stream_out *chunk,
                                        rls_disney_chunk_fn consume /* optional */, void *user);

/* Direct lighting of the rlDisney node: the light loop of shader_evaluate (src/rlDisney.cpp:695-705) --
 *     diffuse  += sampler.evalDiffuseLightSample(sg)     (setSampleType(AI_RAY_DIFFUSE), AiEvaluateLightSample over
 *     specular += sampler.evalSpecularLightSample(sg)     the triple; setSampleType(AI_RAY_GLOSSY), the same: 265-277)
 * = the two direct AOVs (src/rlDisney.cpp:714-715; the indirect_diffuse / indirect_specular factors that secondary
 * rays apply at 707-710 are the caller's).  AiEvaluateLightSample and the light loop are closed: the stand-ins are
 * those of rls_ggx_direct_lighting (spherical lights sampled over the cone they subtend, no occluders, the
 * two-sample power-heuristic estimator with spp_n^2 light samples shared by the two lobes and spp_n^2 BSDF samples
 * per lobe, per light; directions below the shading normal contribute nothing; parity unpinned).  A BSDF sample
 * counts when its pdf exceeds AI_EPSILON, like the valid-sample test of src/rlDisney.cpp:309. */
rls_status rls_disney_direct_lighting(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 P,
                                      const rls_sphere_light *lights, int n_lights, int spp_n, uint32_t seed,
                                      uint64_t first_index, rls_rgb direct_diffuse, rls_rgb direct_specular);

/* shader_evaluate of the rlDisney node for a camera ray, whole (src/rlDisney.cpp:685-727): the light loop (695-705)
 * and integrateDiffuse + integrateGlossy (718-719 -> AiBRDFIntegrate over the triple with the sample type set,
 * 240-243, 279-283) on one closure set-up;
 *     sg->out.RGB = (diffuse + specular) + (indirectDiffuse + indirectGlossy)                    (712, 722)
 * The closed services as in rls_disney_direct_lighting and rls_disney_integrate: AiBRDFIntegrate -> the sum of
 * evalBrdf / evalPdf over the valid samples (pdf > AI_EPSILON, 309) x 1 / spp_n^2 x `env`.  Sample streams: light l
 * 3 l .. 3 l + 2, indirect diffuse 24, indirect glossy 25. */
typedef struct rls_disney_shade_out {
    rls_rgb direct_diffuse, direct_specular, indirect_diffuse, indirect_specular;   /* the AOVs, 714-715, 723-724 */
    rls_rgb out;                                                                    /* optional: sg->out.RGB     */
} rls_disney_shade_out;
rls_status rls_disney_shade(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 P,
                            const rls_sphere_light *lights /* NULL when n_lights == 0 */, int n_lights,
                            const float env[3], int spp_n, uint32_t seed, uint64_t first_index,
                            const rls_disney_shade_out *out);

/* Alternates the reference compiles but never selects (mSampleFromVisibleNormal is hard-wired to
 * true, src/rlDisney.cpp:191): the plain-NDF microfacet samplers, the matching pdf branch and D_GTR2. */
#define RLS_DISNEY_ALT_GTR2_ANISO 0   /* sampleGTR2AnisoDirection, src/rlDisney.cpp:406-414 -> microfacet normal */
#define RLS_DISNEY_ALT_GTR2       1   /* sampleGTR2Direction, src/rlDisney.cpp:504-512 -> direction, not normalised */
rls_status rls_disney_alt_sample(rls_context *ctx, int64_t n, const rls_disney_closure *c, int kind,
                                 const float *rx, const float *ry, rls_vec3 m);
/* evalSpecularPdf with mSampleFromVisibleNormal == false (src/rlDisney.cpp:541-542) */
rls_status rls_disney_alt_pdf(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 wi, float *pdf);
/* D_GTR2 (src/rlDisney.cpp:553-559) */
rls_status rls_disney_d_gtr2(rls_context *ctx, int64_t n, const rls_disney_closure *c, rls_cvec3 m, float *d);

/* ------------------------------------------------------------------------------------------
 * rlSss: NDProfile + SssSampler<NDProfile> hot parts (src/rlSss.h:27-61,143-167,246-266,
 * 401-413,487-545; src/rlSss.cpp:20-106).  Parameter names: src/rlSkin.cpp:109-115.
 * ---------------------------------------------------------------------------------------- */
typedef struct rls_sss_closure {
    rls_param_rgb sss_color;            /* albedo (does not shape the profile: src/rlSss.cpp:23)  */
    rls_param     sss_dist_multiplier;  /* scatterDist = sss_scatter_dist * multiplier            */
    rls_param     sss_scatter_dist[3];  /* VEC parameter: x, y, z                                 */
    rls_cvec3     N;                    /* sg->Ns; may be all-NULL for the profile-only calls     */
    rls_cvec3     T;                    /* sg->dPdu (has_dPdu) or the polar-frame tangent         */
    int           has_dPdu;             /* 1: Gram-Schmidt frame from dPdu (src/rlSss.h:151-154)  */
    rls_material_index materials;       /* optional: the three parameters as per-material columns */
} rls_sss_closure;

/* getRadius(rx) -> r, getPdf(r), evalProfile(r) in one pass (src/rlSss.cpp:36-106) */
rls_status rls_nd_sample(rls_context *ctx, int64_t n, const rls_sss_closure *c, const float *rx,
                         float *r, float *pdf, rls_rgb profile);
rls_status rls_nd_pdf(rls_context *ctx, int64_t n, const rls_sss_closure *c, const float *r, float *pdf);
rls_status rls_nd_eval(rls_context *ctx, int64_t n, const rls_sss_closure *c, const float *r, rls_rgb profile);
/* GaussianProfile (src/rlSss.h:63-97), the alternate profile the reference leaves commented out
 * (src/rlSkin.cpp:242): setDistance(dist.x) then getRadius(rx), getPdf(r), evalProfile(r).  Arnold's
 * closed fast_exp is replaced by exp (parity unpinned). */
rls_status rls_gaussian_sample(rls_context *ctx, int64_t n, rls_param dist_x, const float *rx,
                               float *r, float *pdf, float *profile);
/* getProbeRay (src/rlSss.h:487-533): offset = ray.origin - sg->P, dir, maxdist, r; plus pdf(r)
 * and profile(r).  P (optional, all three planes or none) is added to the offset. */
rls_status rls_sss_probe_ray(rls_context *ctx, int64_t n, const rls_sss_closure *c,
                             const float *rx, const float *ry, rls_cvec3 P,
                             float *r, rls_vec3 origin, rls_vec3 dir, float *maxdist,
                             float *pdf, rls_rgb profile);
/* 3-axis MIS pdf of a probe hit (src/rlSss.h:246-266).  literal_matrix: 0 = projection onto the
 * frame axes (intent), 1 = literal row-vector product ("closed Arnold services"). */
rls_status rls_sss_mis_pdf(rls_context *ctx, int64_t n, const rls_sss_closure *c,
                           rls_cvec3 disp, rls_cvec3 sampleN, int literal_matrix, float *pdf);
/* cavity fade (src/rlSss.h:401-413); r = |disp| (src/rlSss.h:382) */
rls_status rls_sss_cavity_fade(rls_context *ctx, int64_t n, rls_cvec3 disp, rls_cvec3 sampleN,
                               rls_cvec3 No, float *fade);
/* cosine-hemisphere direction about an arbitrary normal (src/rlSss.h:536-545) */
rls_status rls_sss_sample_diffuse_direction(rls_context *ctx, int64_t n, rls_cvec3 normal, rls_cvec3 T,
                                            const float *rx, const float *ry, rls_vec3 wi);

/* SssSampler::integrateScatter (src/rlSss.h:167-280): spp_n^2 probe rays per shading point
 * (getProbeRay, 487-533), each traced through the scene (traceProbe, 293-356), every hit shaded
 * (shadeProbeSample, 361-424: radius cut-off, cavity fade, evalLightSample x evalProfile), the
 * samples combined with the three-axis MIS pdf (246-268), result = sss_color * sum / spp.
 * What the reference obtains from the closed renderer is supplied by an analytic scene
 * (parity unpinned): AiTraceProbe -> ray/plane or ray/sphere intersection (at most two hits;
 * kMaxProbeDepth = 12 is never reached); AiLights* + AiEvaluateLightSample(AiOrenNayarMIS*, sigma 0)
 * -> one distant light on a Lambertian surface, E = light_color / pi * max(0, N.L), optionally lit
 * only where dot(P - gate_point, gate_normal) > 0 (a light/shadow edge, as in the reference's
 * "diffusion decay" test 0010); integrateDiffuse (456-484) -> 0 (shouldTraceDiffuse false).
 * Samples: the per-point scrambled (0,2)-sequence of rls_ggx_integrate. */
#define RLS_SCENE_PLANE  0
#define RLS_SCENE_SPHERE 1
typedef struct rls_sss_scene {
    int   geometry;
    float plane_point[3], plane_normal[3];      /* unit normal, pointing out of the medium         */
    float sphere_center[3], sphere_radius;
    float light_dir[3], light_color[3];         /* unit vector towards the light; radiance         */
    int   has_gate;
    float gate_point[3], gate_normal[3];
    int   use_cavity_fade;                      /* data->useCavityFade(), src/rlSss.h:401          */
    int   literal_matrix;                       /* see rls_sss_mis_pdf                              */
} rls_sss_scene;
/* P: sg->P per shading point; result: integrateScatter's return value; mean_depth (optional):
 * shaded probe hits per probe ray (msgData->probeDepth averaged over the samples). */
rls_status rls_sss_integrate_scatter(rls_context *ctx, int64_t n, const rls_sss_closure *c, rls_cvec3 P,
                                     const rls_sss_scene *scene, int spp_n, uint32_t seed, uint64_t first_index,
                                     rls_rgb result, float *mean_depth);

/* ------------------------------------------------------------------------------------------
 * rlSkin composite: sheen GGX + specular GGX + NDProfile SSS with the layer-weight arithmetic
 * of shader_evaluate (src/rlSkin.cpp:174-246).  Parameter names: src/rlSkin.cpp:109-128.
 * ---------------------------------------------------------------------------------------- */
typedef struct rls_skin_closure {
    rls_cvec3     wo, N, T;             /* T doubles as sg->dPdu for the SSS frame              */
    rls_param_rgb sss_color;
    rls_param     sss_weight, sss_dist_multiplier;
    rls_param     sss_scatter_dist[3];
    rls_param_rgb specular_color;
    rls_param     specular_weight, specular_roughness, specular_ior;
    rls_param_rgb sheen_color;
    rls_param     sheen_weight, sheen_roughness, sheen_ior;
    rls_material_index materials;       /* optional: the parameters above as per-material columns */
} rls_skin_closure;

typedef struct rls_skin_out {
    rls_vec3 sheen_wi;  rls_rgb sheen_f;  float *sheen_pdf;  float *sheen_fresnel;
    rls_vec3 spec_wi;   rls_rgb spec_f;   float *spec_pdf;   float *spec_fresnel;
    float   *r, *r_pdf; rls_rgb profile;
    float   *sheenFresnel, *specularFresnel, *sssWeight;   /* src/rlSkin.cpp:204,228,238 */
} rls_skin_out;

/* xi: six planes {sheen rx, ry, specular rx, ry, sss rx, ry} */
rls_status rls_skin_sample_eval_pdf(rls_context *ctx, int64_t n, const rls_skin_closure *c,
                                    const float *const xi[6], const rls_skin_out *out);

/* shader_evaluate of rlSkin over spp_n^2 samples per layer (src/rlSkin.cpp:174-254).  Per GGX lobe (sheen, then
 * specular; skipped when its weight <= AI_EPSILON, 191/214) integrateGlossy's sample loop; every evalSample of it
 * adds its Fresnel term to the closure (src/rlGgx.h:103) and getAvgReflectWeight (181-184) -- the mean over ALL the
 * samples of the lobe, 1 when none were drawn -- is what the layer hands down:
 *     sheenFresnel    = avg_sheen * sheen_weight                      (204)    sheen    *= sheen_weight        (207)
 *     specularFresnel = avg_spec  * specular_weight                   (228)    specular *= specular_weight * (1 - sheenFresnel)   (231)
 *     sssWeight       = sss_weight * (1 - specularFresnel * (1 - sheenFresnel))   (238)
 *     sss = sssWeight < AI_EPSILON ? black : integrateScatter(scatterDist = sss_scatter_dist * sss_dist_multiplier) * sssWeight   (244-246)
 * The mean is a wave-shuffle reduction over the lanes that share a shading point (a17).  What the reference gets
 * from the closed renderer is supplied as for the single-closure integrators (parity unpinned): AiBRDFIntegrate ->
 * the mean of evalBrdf / evalPdf over the samples times the radiance env[3] of a uniform environment;
 * integrateScatter -> rls_sss_integrate_scatter's analytic scene; the light loops of 193-198 / 217-222
 * (evalLightSample per light, BEFORE integrateGlossy) -> the estimator of rls_ggx_direct_lighting's specular lobe
 * over `lights` (n_lights = 0: no lights, the loops draw no samples; here the two strategies' terms go into ONE sum,
 * sample by sample).  The BSDF-sampling half of that estimator calls
 * evalSample, so its Fresnel terms enter the mean too: getAvgReflectWeight = (sum over the light loops' BSDF samples
 * and integrateGlossy's samples) / (their count).  integrateGlossy draws no samples for a small colour (174-176),
 * the light loop does (167-170).  Scramble streams: sheen glossy 0, specular glossy 1, scatter 2 (as without
 * lights), then light l of the sheen lobe 3 + 4 l (light samples), 4 + 4 l (BSDF samples), of the specular lobe
 * 5 + 4 l, 6 + 4 l. */
typedef struct rls_skin_integrate_out {
    rls_rgb sheen, specular, sss;                          /* the three AOVs (src/rlSkin.cpp:249-251)            */
    rls_rgb out;                                           /* optional: sg->out.RGB = their sum (254)            */
    float  *sheenFresnel, *specularFresnel, *sssWeight;    /* optional: the hand-down scalars (204, 228, 238)    */
} rls_skin_integrate_out;
rls_status rls_skin_integrate(rls_context *ctx, int64_t n, const rls_skin_closure *c, rls_cvec3 P,
                              const rls_sss_scene *scene, const float env[3],
                              const rls_sphere_light *lights /* NULL when n_lights == 0 */, int n_lights,
                              int spp_n, uint32_t seed, uint64_t first_index, const rls_skin_integrate_out *out);

/* ------------------------------------------------------------------------------------------
 * rlUtil closures (src/rlUtil.h:21-29, src/rlUtil.cpp:3-27), batch form for parity checks:
 * spherical = sphericalDirection(2a-1, 2*pi*b), disk = concentricDiskSample(a, b) (z = 0).
 * ---------------------------------------------------------------------------------------- */
rls_status rls_util_directions(rls_context *ctx, int64_t n, const float *a, const float *b,
                               rls_vec3 spherical, rls_vec3 disk);
/* reflected = reflectDirection(i, nrm) = 2 |i.nrm| nrm - i (src/rlUtil.h:31-34; note the ABS),
 * luminance = colorToLuminance(color) (src/rlUtil.h:36-39) */
rls_status rls_util_reflect_luminance(rls_context *ctx, int64_t n, rls_cvec3 i, rls_cvec3 nrm, rls_cvec3 color,
                                      rls_vec3 reflected, float *luminance);

/* ------------------------------------------------------------------------------------------
 * Synthetic shading-point generator (bench / tests): counter-based, value = f(seed, index,
 * stream); integer hashing and + - * / sqrt only, so the CPU oracle's generator matches bit
 * for bit.  Distributions: DESIGN.md "Synthetic inputs".
 * ---------------------------------------------------------------------------------------- */
rls_status rls_gen_frame(rls_context *ctx, uint32_t seed, uint64_t first_index, int64_t n,
                         rls_vec3 wo, rls_vec3 N, rls_vec3 T);
rls_status rls_gen_uniform(rls_context *ctx, uint32_t seed, uint64_t first_index, int64_t n,
                           uint32_t stream, float lo, float hi, float *out);
rls_status rls_gen_aniso(rls_context *ctx, uint32_t seed, uint64_t first_index, int64_t n, float *out);
/* order-independent 64-bit checksum of n floats (sum of per-element hashes of the bit patterns) */
rls_status rls_checksum(rls_context *ctx, int64_t n, const float *data, uint64_t *out_host);

/* Validation only: out[i] = one elementary function of the context's arithmetic mode, evaluated on the device
 * by the routines the closure kernels use (RLS_MATH_EXACT: the host-libm-faithful ones; RLS_MATH_FAST: the
 * hardware ones).  tools/libm_exhaustive.py sweeps all 2^32 arguments of the unary functions through it. */
#define RLS_FN_SQRT  0   /* sqrtf(x)     */
#define RLS_FN_DIV   1   /* x / y        */
#define RLS_FN_ATAN2 2   /* atan2f(x, y) */
#define RLS_FN_ACOS  3
#define RLS_FN_TAN   4
#define RLS_FN_SIN   5
#define RLS_FN_COS   6
#define RLS_FN_EXP   7
#define RLS_FN_LOG   8
#define RLS_FN_POW   9   /* powf(x, y)   */
/* the forms the closure kernels call for their angles, which are bounded by construction (results of atan2f /
 * acosf, or 2 pi xi): identical to RLS_FN_TAN / _SIN / _COS for |x| < 120 and for NaN, unspecified beyond */
#define RLS_FN_TAN_BOUNDED 10
#define RLS_FN_SIN_BOUNDED 11
#define RLS_FN_COS_BOUNDED 12
rls_status rls_libm_eval(rls_context *ctx, int fn, int64_t n, const float *x, const float *y, float *out);

#ifdef __cplusplus
}
#endif
#endif /* RLSHADERS_AMD_H */
