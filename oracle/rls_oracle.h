/*
 * rls_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, scalar, one-point-at-a-time restatement of the closure arithmetic of
 * shihchinw/rlShaders (reference paths cited per function, relative to /root/reference).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (rlshaders_amd/, include/rlshaders_amd.h) never links, imports or calls it.
 *
 * PARITY STATUS: **parity unpinned**.
 *   - The reference cannot be built in this image: every TU includes the closed
 *     Arnold SDK header <ai.h> (src/rlUtil.h:9, src/rlGgx.h:17) which is absent, and
 *     no stand-in for it is written (task rule).  There is no oracle/_ref.
 *   - The reference's own tests are 10 whole-image renders that need Arnold `kick`
 *     (testsuite/runtest.py:193-244); there are no function-level golden vectors.
 *   - The only numbers that trace back to an execution of the reference are the
 *     probe known-answer values recorded in SURVEY.md section 8(c); this restatement
 *     reproduces them (tests/test_oracle_kat.py, tests/golden/survey_kat.json).
 *   - Arnold inline helpers are restated from their public 4.x definitions
 *     (SURVEY.md Appendix C): AiV3Normalize = multiply by 1/len (0 if len==0),
 *     LERP(t,a,b) = (1-t)*a + b*t, LINEARSTEP = CLAMP((t-lo)/(hi-lo),0,1),
 *     SGN(a) = a<0 ? -1 : 1, AiV3RotateToFrame(a,u,v,w) = a.x*u + a.y*v + a.z*w.
 *   - Closed Arnold services are replaced at the boundary: AiBuildLocalFramePolar ->
 *     the tangent T is an INPUT (V = N x T); AiRefractRay/AiReflectRay -> Snell /
 *     mirror (Walter et al. EGSR'07 eq. 40); AiM4Frame+AiM4VectorByMatrixMult ->
 *     projection onto (U,V,N) (intent) or the literal row-vector form (flag);
 *     AiSampler -> xi is an input.
 *   - Unqualified exp/log/pow/sqrt in the reference (src/rlSss.cpp:31-32,59,62,78-79,
 *     102; src/rlDisney.cpp:177,399,549,576; src/rlGgx.h:148) are taken as the FLOAT
 *     overloads (MSVC, the author's platform; libstdc++ with <math.h>): fp32 throughout.
 *
 * Build: oracle/Makefile -> oracle/build/librls_oracle.so  (gcc -O2 -ffp-contract=off,
 * no fast-math: the reference sets no -O/fast-math flags, CMakeLists.txt:58).
 */
#ifndef RLS_ORACLE_H
#define RLS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_v3;
typedef struct { float x, y; } orc_v2;
typedef struct { float r, g, b; } orc_rgb;

/* Arnold ray-type tags used by DisneySampler::mSampleType (src/rlDisney.cpp:112,132,147) */
#define ORC_RAY_DIFFUSE 0x08
#define ORC_RAY_GLOSSY  0x10

/* ---- rlUtil (src/rlUtil.h:21-39, src/rlUtil.cpp:3-27) ------------------------------- */
orc_v3 orc_spherical_direction(float cosTheta, float phi);
orc_v3 orc_reflect_direction(orc_v3 i, orc_v3 n);
float  orc_color_to_luminance(orc_rgb c);
orc_v2 orc_concentric_disk_sample(float rx, float ry);

/* ---- rlGgx: GgxSamplerT<VNDFKernel> (src/rlGgx.h:92-373, src/rlGgx.cpp:14-99) ------ */
typedef struct {
    orc_v3  U, V, N;          /* mBasis                              */
    orc_v3  axisN;            /* mAxisN                              */
    orc_v3  viewDir;          /* mViewDir = -sg->Rd                  */
    orc_rgb specColor;        /* mSpecularColor                      */
    float   roughness;        /* mRoughness = max(1e-5, r^2)         */
    float   alphaX, alphaY;
    float   iorIn, iorOut;
    float   reflectWeight;    /* mutable mReflectWeight              */
    float   misSampleCount;   /* mutable mMisSampleCount             */
} orc_ggx;

void    orc_ggx_init(orc_ggx *g, orc_v3 wo, orc_v3 Nf, orc_v3 T, int exiting,
                     orc_rgb specColor, float ior, float roughness, float anisotropic);
orc_v2  orc_vndf_sample_slope(float theta, float rx, float ry);
orc_v3  orc_vndf_sample(const orc_ggx *g, float rx, float ry);            /* microfacet normal M */
orc_v3  orc_ndf_sample(const orc_ggx *g, float rx, float ry);             /* NDFKernel alt.      */
float   orc_ggx_fresnel(const orc_ggx *g, orc_v3 i, orc_v3 m);
float   orc_ggx_D(const orc_ggx *g, orc_v3 m);
float   orc_ggx_G1(const orc_ggx *g, orc_v3 v, orc_v3 m, orc_v3 n);
float   orc_ggx_G(const orc_ggx *g, orc_v3 i, orc_v3 o, orc_v3 m, orc_v3 n);
float   orc_ggx_reflection(const orc_ggx *g, orc_v3 i, orc_v3 o, orc_v3 n);
float   orc_ggx_refraction(const orc_ggx *g, orc_v3 i, orc_v3 o, orc_v3 n);  /* dead code in ref */
float   orc_ggx_sample_weight(const orc_ggx *g, orc_v3 i, orc_v3 o, orc_v3 m);
float   orc_vndf_pdf(const orc_ggx *g, orc_v3 i, orc_v3 m);
float   orc_ndf_pdf(const orc_ggx *g, orc_v3 i, orc_v3 m);
/* the static callback triple (src/rlGgx.h:97-127) */
orc_v3  orc_ggx_eval_sample(orc_ggx *g, float rx, float ry);               /* accumulates Fresnel */
orc_rgb orc_ggx_eval_brdf(const orc_ggx *g, orc_v3 indir);
float   orc_ggx_eval_pdf(const orc_ggx *g, orc_v3 indir);
float   orc_ggx_avg_reflect_weight(const orc_ggx *g);
/* per-sample body of integrateRefract (src/rlGgx.h:228-242); returns 1 if refracted, 0 on TIR */
int     orc_ggx_refract_sample(const orc_ggx *g, float rx, float ry, orc_v3 *dir, float *weight);

/* ---- rlDisney: DisneySampler (src/rlDisney.cpp:105-602) ----------------------------- */
typedef struct {
    orc_v3  viewDir, axisU, axisV, axisN;
    orc_rgb specularF0, sheenColor, baseColor;
    float   roughness, subsurface, specular, specularTint, metallic, sheen, sheenTint,
            anisotropic, clearcoat, clearcoatGloss;
    float   specularRoughness, alphaX, alphaY;
    int     sampleType;               /* mSampleType                */
    int     sampleFromVisibleNormal;  /* mSampleFromVisibleNormal   */
} orc_disney;

/* scalars in rlDisney parameter order (src/rlDisney.cpp:608-610):
   subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint,
   clearcoat, clearcoat_gloss */
void    orc_disney_init(orc_disney *d, orc_v3 wo, orc_v3 Nf, orc_v3 T, orc_rgb base_color,
                        const float scalars[10]);
orc_rgb orc_disney_eval_diffuse(const orc_disney *d, orc_v3 L);
orc_rgb orc_disney_eval_specular(const orc_disney *d, orc_v3 L);
orc_v3  orc_disney_sample_diffuse(const orc_disney *d, float rx, float ry);
orc_v3  orc_disney_sample_specular(const orc_disney *d, float rx, float ry);
float   orc_disney_diffuse_pdf(const orc_disney *d, orc_v3 i);
float   orc_disney_specular_pdf(const orc_disney *d, orc_v3 i);
orc_v3  orc_disney_eval_sample(const orc_disney *d, float rx, float ry);
orc_rgb orc_disney_eval_brdf(const orc_disney *d, orc_v3 indir);
float   orc_disney_eval_pdf(const orc_disney *d, orc_v3 indir);

/* alternates the reference compiles but never selects (mSampleFromVisibleNormal = true,
 * src/rlDisney.cpp:191): sampleGTR2AnisoDirection (406-414), sampleGTR2Direction (504-512, returns an
 * un-normalised direction), D_GTR2 (553-559); the non-VNDF branch of evalSpecularPdf (541-542) is
 * reached by clearing d->sampleFromVisibleNormal */
orc_v3  orc_disney_sample_gtr2_aniso(const orc_disney *d, float rx, float ry);
orc_v3  orc_disney_sample_gtr2(const orc_disney *d, float rx, float ry);
float   orc_disney_D_GTR2(const orc_disney *d, orc_v3 m);
/* scalar access to the file-local helpers smithG_GGX, D_GTR1, D_GTR2Aniso (src/rlDisney.cpp:545-577) for KATs */
float   orc_kat_smithG_GGX(float NdotV, float alphaG);
float   orc_kat_D_GTR1(float clearcoat_gloss, float MdotN2);
float   orc_kat_D_GTR2Aniso(float alphaX, float alphaY, orc_v3 m);

/* GaussianProfile (src/rlSss.h:63-97; not instantiated: src/rlSkin.cpp:242 is commented out).  Arnold's
 * closed fast_exp is replaced by expf: PARITY UNPINNED. */
typedef struct { float variance, maxRadius, norm; } orc_gauss;
void    orc_gauss_set_distance(orc_gauss *g, float dist_x);
float   orc_gauss_get_radius(const orc_gauss *g, float rx);
float   orc_gauss_get_pdf(const orc_gauss *g, float r);
float   orc_gauss_eval_profile(const orc_gauss *g, float r);
void    orc_batch_gauss(int64_t n, const float *dist_x, const float *rx, float *r, float *pdf, float *profile,
                        int nthreads);

/* ---- rlSss: NDProfile (src/rlSss.h:27-61, src/rlSss.cpp:20-106) --------------------- */
typedef struct {
    float distance[3];
    float c1[3], c2[3];
    float maxRadius;
} orc_nd;

void    orc_nd_set_distance(orc_nd *p, orc_v3 dist, orc_rgb albedo);
int     orc_nd_select_dist_lobe(float *x);
float   orc_nd_get_radius(const orc_nd *p, float rx);
float   orc_nd_get_pdf(const orc_nd *p, float r);
orc_rgb orc_nd_eval_profile(const orc_nd *p, float r);

/* ---- rlSss: SssSampler<NDProfile> hot parts (src/rlSss.h:143-167,246-266,401-413,487-545) */
typedef struct {
    orc_nd  profile;
    orc_rgb baseColor;
    orc_v3  axisU, axisV, axisN;
} orc_sss;

/* has_dPdu: 1 -> Gram-Schmidt from dPdu (src/rlSss.h:151-154); 0 -> T used as the polar-frame
   tangent (closed AiBuildLocalFramePolar replaced: U = T, V = N x T) */
void    orc_sss_init(orc_sss *s, orc_v3 Ns, orc_v3 dPdu_or_T, int has_dPdu, orc_rgb albedo, orc_v3 dist);
/* returns r; offset is ray.origin - sg->P */
float   orc_sss_get_probe_ray(const orc_sss *s, float rx, float ry,
                              orc_v3 *offset, orc_v3 *dir, float *maxdist);
/* literal_matrix: 0 -> projection (disp.U, disp.V, disp.N); 1 -> literal row-vector product
   disp.x*U + disp.y*V + disp.z*N */
float   orc_sss_mis_pdf(const orc_sss *s, orc_v3 disp, orc_v3 sampleN, int literal_matrix);
float   orc_sss_cavity_fade(orc_v3 disp, float r, orc_v3 sampleN, orc_v3 No);
orc_v3  orc_sss_sample_diffuse_direction(float rx, float ry, orc_v3 normal, orc_v3 T);

/* ---- rlSkin layer arithmetic (src/rlSkin.cpp:174-246) ------------------------------- */
typedef struct {
    orc_rgb sss_color;  float sss_weight; float sss_dist_multiplier; orc_v3 sss_scatter_dist;
    orc_rgb specular_color; float specular_weight, specular_roughness, specular_ior;
    orc_rgb sheen_color;    float sheen_weight, sheen_roughness, sheen_ior;
} orc_skin_params;

typedef struct {
    /* per GGX lobe: one (sample, eval, pdf) triple + the Fresnel term it accumulates */
    orc_v3  sheen_wi;  orc_rgb sheen_f;  float sheen_pdf;  float sheen_fresnel;
    orc_v3  spec_wi;   orc_rgb spec_f;   float spec_pdf;   float spec_fresnel;
    /* SSS: radius sample, its pdf, the profile */
    float   r, r_pdf;  orc_rgb profile;
    /* layer weights: sheenFresnel, specularFresnel, sssWeight' (src/rlSkin.cpp:204,228,238),
       spec_scale = specular_weight*(1-sheenFresnel) (src/rlSkin.cpp:231) */
    float   sheenFresnel, specularFresnel, sssWeight, specScale;
} orc_skin_out;

/* xi: {sheen rx, ry, spec rx, ry, sss rx, ry} */
void    orc_skin_eval(const orc_skin_params *p, orc_v3 wo, orc_v3 Nf, orc_v3 T,
                      const float xi[6], orc_skin_out *out);

/* =====================================================================================
 * Batch drivers over planar SoA arrays (what tests / bench call through ctypes).
 * Every array is n floats; vec3/rgb are three separate planes.  NULL optional planes
 * are documented per function.  nthreads<=1 runs inline; >1 uses pthreads, static chunks.
 * ===================================================================================== */
typedef struct { const float *x, *y, *z; } orc_cv3p;
typedef struct { float *x, *y, *z; } orc_v3p;

typedef struct {
    orc_cv3p wo, N, T;
    orc_cv3p KsColor;
    const float *specularRoughness, *ior, *anisotropic;   /* anisotropic may be NULL (=0) */
    const uint8_t *exiting;                                /* may be NULL (=entering)      */
} orc_ggx_soa;

/* fused reflect triple: wi, f, pdf, fresnel(L,M) */
void orc_batch_ggx_sample_eval_pdf(int64_t n, const orc_ggx_soa *in, const float *rx, const float *ry,
                                   orc_v3p wi, orc_v3p f, float *pdf, float *fresnel, int nthreads);
void orc_batch_ggx_sample(int64_t n, const orc_ggx_soa *in, const float *rx, const float *ry,
                          orc_v3p wi, float *fresnel, int nthreads);
void orc_batch_ggx_eval(int64_t n, const orc_ggx_soa *in, orc_cv3p wi, orc_v3p f, int nthreads);
void orc_batch_ggx_pdf(int64_t n, const orc_ggx_soa *in, orc_cv3p wi, float *pdf, int nthreads);
void orc_batch_ggx_refract(int64_t n, const orc_ggx_soa *in, const float *rx, const float *ry,
                           orc_v3p wt, float *weight, uint8_t *refracted, int nthreads);
/* reflect triple with (rx,ry) + refract sample with (rx2,ry2) on ONE closure object per point,
 * as shader_evaluate does (src/rlGgx.cpp:261,293,303); this is the timed CPU baseline */
void orc_batch_ggx_reflect_refract(int64_t n, const orc_ggx_soa *in, const float *rx, const float *ry,
                                   const float *rx2, const float *ry2,
                                   orc_v3p wi, orc_v3p f, float *pdf, float *fresnel,
                                   orc_v3p wt, float *weight, int nthreads);
/* microfacet normal only (decoupled check of a7/a8) */
void orc_batch_ggx_microfacet(int64_t n, const orc_ggx_soa *in, const float *rx, const float *ry,
                              orc_v3p m, int use_ndf_kernel, int nthreads);
void orc_batch_ggx_ndf_pdf(int64_t n, const orc_ggx_soa *in, orc_cv3p wi, float *pdf, int nthreads);

typedef struct {
    orc_cv3p wo, N, T;
    orc_cv3p base_color;
    const float *scalars[10];   /* rlDisney order, see orc_disney_init */
} orc_disney_soa;

void orc_batch_disney_sample(int64_t n, const orc_disney_soa *in, int lobe, const float *rx, const float *ry,
                             orc_v3p wi, int nthreads);
void orc_batch_disney_eval(int64_t n, const orc_disney_soa *in, int lobe, orc_cv3p wi, orc_v3p f, int nthreads);
void orc_batch_disney_pdf(int64_t n, const orc_disney_soa *in, int lobe, orc_cv3p wi, float *pdf, int nthreads);
void orc_batch_disney_sample_eval_pdf(int64_t n, const orc_disney_soa *in, int lobe,
                                      const float *rx, const float *ry,
                                      orc_v3p wi, orc_v3p f, float *pdf, int nthreads);

/* kind 0: sampleGTR2AnisoDirection(rx,ry) -> out3, 1: sampleGTR2Direction -> out3,
   2: non-VNDF evalSpecularPdf(v) -> out1, 3: D_GTR2(v) -> out1 */
void orc_batch_disney_alt(int64_t n, const orc_disney_soa *in, int kind, const float *rx, const float *ry,
                          orc_cv3p v, orc_v3p out3, float *out1, int nthreads);

typedef struct {
    orc_cv3p sss_scatter_dist;   /* already multiplied by sss_dist_multiplier unless mult given */
    const float *sss_dist_multiplier; /* may be NULL (=1, no multiply performed) */
    orc_cv3p sss_color;
    orc_cv3p N, T;               /* Ns and dPdu/T; may be all-NULL for profile-only calls */
} orc_sss_soa;

void orc_batch_nd_sample_pdf_profile(int64_t n, const orc_sss_soa *in, const float *rx,
                                     float *r, float *pdf, orc_v3p profile, int nthreads);
void orc_batch_nd_pdf(int64_t n, const orc_sss_soa *in, const float *r, float *pdf, int nthreads);
void orc_batch_nd_profile(int64_t n, const orc_sss_soa *in, const float *r, orc_v3p profile, int nthreads);
void orc_batch_sss_probe(int64_t n, const orc_sss_soa *in, int has_dPdu, const float *rx, const float *ry,
                         float *r, orc_v3p offset, orc_v3p dir, float *maxdist,
                         float *pdf, orc_v3p profile, int nthreads);
void orc_batch_sss_mis_pdf(int64_t n, const orc_sss_soa *in, int has_dPdu, orc_cv3p disp, orc_cv3p sampleN,
                           int literal_matrix, float *pdf, int nthreads);
void orc_batch_sss_cavity_fade(int64_t n, orc_cv3p disp, orc_cv3p sampleN, orc_cv3p No, float *fade, int nthreads);
void orc_batch_sss_sample_diffuse(int64_t n, orc_cv3p normal, orc_cv3p T, const float *rx, const float *ry,
                                  orc_v3p wi, int nthreads);

typedef struct {
    orc_cv3p wo, N, T;
    orc_cv3p sss_color; const float *sss_weight; const float *sss_dist_multiplier; orc_cv3p sss_scatter_dist;
    orc_cv3p specular_color; const float *specular_weight, *specular_roughness, *specular_ior;
    orc_cv3p sheen_color;    const float *sheen_weight, *sheen_roughness, *sheen_ior;
    const float *xi[6];
} orc_skin_soa;

typedef struct {
    orc_v3p sheen_wi, sheen_f; float *sheen_pdf, *sheen_fresnel;
    orc_v3p spec_wi,  spec_f;  float *spec_pdf,  *spec_fresnel;
    float *r, *r_pdf; orc_v3p profile;
    float *sheenFresnel, *specularFresnel, *sssWeight;
} orc_skin_out_soa;

void orc_batch_skin(int64_t n, const orc_skin_soa *in, const orc_skin_out_soa *out, int nthreads);

/* n^2-spp integrators: the per-sample loop arithmetic Arnold's AiBRDFIntegrate runs over the
 * callback triple (explicit form: src/rlDisney.cpp:299-312), samples from the per-point scrambled
 * (0,2)-sequence below (stand-in for the closed AiSampler(n,2)); sums in ascending sample order. */
void orc_sample_02(uint32_t seed, uint64_t index, uint32_t dim_pair, uint32_t s, float *rx, float *ry);
/* first_index: global index of point 0 (scrambles hash first_index + i) */
void orc_batch_ggx_integrate(int64_t n, const orc_ggx_soa *in, int spp_n, uint32_t seed, uint64_t first_index,
                             orc_v3p sum_f_over_pdf, float *avg_reflect_weight, int nthreads);
/* s_wi/s_f/s_pdf: optional streamed per-sample planes, index lobe*n*spp + s*n + i (lobe 0 = diffuse) */
void orc_batch_disney_integrate(int64_t n, const orc_disney_soa *in, int spp_n, uint32_t seed, uint64_t first_index,
                                orc_v3p dsum, float *dcount, orc_v3p ssum, float *scount,
                                orc_v3p s_wi, orc_v3p s_f, float *s_pdf, int nthreads);

/* SssSampler::integrateScatter (src/rlSss.h:167-280) with traceProbe / shadeProbeSample / evalLightSample
 * (293-356, 361-424, 439-454) over an analytic scene: the closed AiTraceProbe becomes a ray/plane or
 * ray/sphere intersection, the closed AiLights* / AiEvaluateLightSample(AiOrenNayarMIS*, sigma = 0)
 * becomes one distant light on a Lambertian surface (E = light_color / pi * max(0, N.L)), optionally
 * gated by a half-space (lit where dot(P - gate_point, gate_normal) > 0: the light/shadow edge of the
 * reference's "diffusion decay" test scene).  integrateDiffuse (456-484) contributes nothing
 * (shouldTraceDiffuse false).  Same layout as rls_sss_scene in include/rlshaders_amd.h. */
typedef struct {
    int   geometry;                       /* 0: plane, 1: sphere */
    float plane_point[3], plane_normal[3];
    float sphere_center[3], sphere_radius;
    float light_dir[3], light_color[3];
    int   has_gate;
    float gate_point[3], gate_normal[3];
    int   use_cavity_fade;
    int   literal_matrix;
} orc_scene;
/* up to two probe hits of (O, D, maxdist) in ascending t; returns the count */
int  orc_scene_trace(const orc_scene *sc, orc_v3 O, orc_v3 D, float maxdist, float t[2], orc_v3 hitP[2], orc_v3 hitN[2]);
void orc_batch_sss_integrate_scatter(int64_t n, const orc_sss_soa *in, int has_dPdu, orc_cv3p P,
                                     const orc_scene *sc, int spp_n, uint32_t seed, uint64_t first_index,
                                     orc_v3p result, float *mean_depth, int nthreads);

void orc_sss_scatter_point(const orc_sss *S, orc_v3 Po, const orc_scene *sc, int spp, uint32_t seed, uint64_t index,
                           uint32_t dim_pair, float acc[3], float *depth_sum);

/* shader_evaluate of rlSkin over spp samples per layer (src/rlSkin.cpp:174-254): integrateGlossy per GGX lobe with
 * the Fresnel mean of getAvgReflectWeight (src/rlGgx.h:181-184) handed down, integrateScatter x sssWeight.
 * AiBRDFIntegrate -> mean of evalBrdf / evalPdf x env (uniform environment); scene as above.
 * Dimension pairs of the sampler: 0 sheen, 1 specular, 2 probe rays. */
/* a spherical light of the light loops' stand-in (see orc_batch_ggx_direct_lighting); layout of rls_sphere_light */
typedef struct {
    float center[3], radius;
    float radiance[3];
    int   mis_mode;
} orc_light;
typedef struct {
    orc_rgb sheen, specular, sss, out;
    float sheenFresnel, specularFresnel, sssWeight;
} orc_skin_int_out;
/* lights / n_lights: the light loops of src/rlSkin.cpp:193-198, 217-222 (n_lights 0: none); streams 3 + 4 l, 4 + 4 l
 * (sheen lobe: light samples, BSDF samples), 5 + 4 l, 6 + 4 l (specular lobe) */
void orc_skin_integrate(const orc_skin_params *p, orc_v3 wo, orc_v3 Nf, orc_v3 T, orc_v3 P, const orc_scene *sc,
                        const float env[3], const orc_light *lights, int n_lights, int spp, uint32_t seed, uint64_t index,
                        orc_skin_int_out *out);
typedef struct {
    orc_v3p sheen, specular, sss, out;
    float *sheenFresnel, *specularFresnel, *sssWeight;
} orc_skin_int_out_soa;
void orc_batch_skin_integrate(int64_t n, const orc_skin_soa *in, orc_cv3p P, const orc_scene *sc, const float env[3],
                              const orc_light *lights, int n_lights, int spp_n, uint32_t seed, uint64_t first_index,
                              const orc_skin_int_out_soa *out, int nthreads);

/* integrateRefract (src/rlGgx.h:205-245): traced -> the sample loop (228-244), else the single refraction about
 * the shading normal (213-222); radiance of a uniform environment env */
orc_rgb orc_ggx_integrate_refract(const orc_ggx *g, int traced, const float env[3], int spp, uint32_t seed,
                                  uint64_t index, float *tir_fraction);
void orc_batch_ggx_integrate_refract(int64_t n, const orc_ggx_soa *in, int traced, const float env[3], int spp_n,
                                     uint32_t seed, uint64_t first_index, orc_v3p result, float *tir_fraction, int nthreads);

/* Direct lighting of the rlGgx node (shader_evaluate's light loop, src/rlGgx.cpp:274-299): per light
 * sample `diffuse += AiEvaluateLightSample(sg, diffData, AiOrenNayarMIS*)` and `specular +=
 * sampler.evalLightSample(sg)` (= AiEvaluateLightSample over the GGX triple, src/rlGgx.h:167-170),
 * then diffuse *= Kd_color * Kd, specular *= Ks.  Arnold's light loop, AiEvaluateLightSample and the
 * Oren-Nayar MIS closure are closed; documented stand-ins (parity unpinned): one spherical area light
 * sampled uniformly over the cone it subtends, the qualitative Oren-Nayar model (SIGGRAPH'94) with
 * cosine-weighted sampling, and the two-sample estimator with the power heuristic
 * w_a = p_a^2 / (p_a^2 + p_b^2) over spp_n^2 light samples and spp_n^2 BSDF samples per lobe.
 * mis_mode 1 / 2 keep only the light / only the BSDF samples (weight 1): the three modes have the same
 * expectation, which is what the tests check. */
typedef struct {
    orc_cv3p Kd_color;
    const float *Kd, *Kd_roughness, *Ks;
    orc_cv3p Kt_color;          /* read by orc_batch_ggx_shade only */
    const float *Kt;
} orc_ggx_shader_soa;
typedef struct { orc_v3 N, T; float A, B; } orc_oren_nayar;
void  orc_oren_nayar_init(orc_oren_nayar *o, orc_v3 N, orc_v3 T, float sigma);
float orc_oren_nayar_brdf(const orc_oren_nayar *o, orc_v3 wo, orc_v3 wi);   /* BRDF x cos(theta_i) */
float orc_oren_nayar_pdf(const orc_oren_nayar *o, orc_v3 wi);
/* lights: n_lights of them; light l uses the sample streams 3 l (light samples, shared by both lobes), 3 l + 1 and
 * 3 l + 2 (BSDF samples of the specular / diffuse lobe); the AOVs are the sums over the lights in array order */
void  orc_batch_ggx_direct_lighting(int64_t n, const orc_ggx_soa *in, const orc_ggx_shader_soa *sh, orc_cv3p P,
                                    const orc_light *lights, int n_lights, int spp_n, uint32_t seed, uint64_t first_index,
                                    orc_v3p direct_diffuse, orc_v3p direct_specular, int nthreads);
/* Summation order of the light loops' estimator (rls_oracle.c, ggx_light_loop): the entry points above and below keep ONE
 * running sum per AOV, each sample's terms added as they come (light sample, then BSDF sample) -- the canonical form.
 * The *_two_sums twins keep one sum per strategy and add the two at the end: the order the device kernels produce (they
 * run the strategies as separate passes).  Same terms; the forms agree to 1e-6 (tests/test_oracle_light_loops.py). */
void  orc_batch_ggx_direct_lighting_two_sums(int64_t n, const orc_ggx_soa *in, const orc_ggx_shader_soa *sh, orc_cv3p P,
                                             const orc_light *lights, int n_lights, int spp_n, uint32_t seed,
                                             uint64_t first_index, orc_v3p direct_diffuse, orc_v3p direct_specular, int nthreads);
/* shader_evaluate of rlGgx for a camera ray, whole (src/rlGgx.cpp:248-327): the light loop, transmission =
 * integrateRefract x KtColor x Kt (black for a small product), indirect diffuse = diffuseColor x AiBRDFIntegrate over the
 * Oren-Nayar closure (mean of brdf / pdf over cosine-weighted samples x env), indirect glossy = integrateGlossy x Ks,
 * out = (diffuse + specular + transmission) + (indirectDiffuse + indirectGlossy).  Sample streams: light l 3 l .. 3 l + 2,
 * integrateGlossy 24, integrateRefract 25, indirect diffuse 26.  n_lights may be 0; out->out.x may be NULL. */
typedef struct { orc_v3p direct_diffuse, direct_specular, refraction, indirect_diffuse, indirect_specular, out; } orc_ggx_shade_out_soa;
void  orc_batch_ggx_shade(int64_t n, const orc_ggx_soa *in, const orc_ggx_shader_soa *sh, orc_cv3p P,
                          const orc_light *lights, int n_lights, const float env[3], int traced, int spp_n, uint32_t seed,
                          uint64_t first_index, const orc_ggx_shade_out_soa *out, int nthreads);
void  orc_batch_ggx_shade_two_sums(int64_t n, const orc_ggx_soa *in, const orc_ggx_shader_soa *sh, orc_cv3p P,
                                   const orc_light *lights, int n_lights, const float env[3], int traced, int spp_n,
                                   uint32_t seed, uint64_t first_index, const orc_ggx_shade_out_soa *out, int nthreads);
/* shader_evaluate of rlDisney for a camera ray, whole (src/rlDisney.cpp:685-727): the light loop + integrateDiffuse +
 * integrateGlossy (sum of brdf / pdf over the valid samples x 1 / spp x env); streams: lights as above, 24, 25 */
typedef struct { orc_v3p direct_diffuse, direct_specular, indirect_diffuse, indirect_specular, out; } orc_disney_shade_out_soa;
void  orc_batch_disney_shade(int64_t n, const orc_disney_soa *in, orc_cv3p P, const orc_light *lights, int n_lights,
                             const float env[3], int spp_n, uint32_t seed, uint64_t first_index,
                             const orc_disney_shade_out_soa *out, int nthreads);
void  orc_batch_disney_shade_two_sums(int64_t n, const orc_disney_soa *in, orc_cv3p P, const orc_light *lights, int n_lights,
                                      const float env[3], int spp_n, uint32_t seed, uint64_t first_index,
                                      const orc_disney_shade_out_soa *out, int nthreads);
/* Direct lighting of the rlDisney node (src/rlDisney.cpp:695-705: evalDiffuseLightSample + evalSpecularLightSample per
 * light, 265-277), same stand-ins; streams 3 l (light samples), 3 l + 1 (diffuse BSDF samples), 3 l + 2 (specular) */
void  orc_batch_disney_direct_lighting(int64_t n, const orc_disney_soa *in, orc_cv3p P, const orc_light *lights,
                                       int n_lights, int spp_n, uint32_t seed, uint64_t first_index,
                                       orc_v3p direct_diffuse, orc_v3p direct_specular, int nthreads);
void  orc_batch_disney_direct_lighting_two_sums(int64_t n, const orc_disney_soa *in, orc_cv3p P, const orc_light *lights,
                                                int n_lights, int spp_n, uint32_t seed, uint64_t first_index,
                                                orc_v3p direct_diffuse, orc_v3p direct_specular, int nthreads);

/* utility closures, batch form (a2-a5) */
void orc_batch_util(int64_t n, const float *a, const float *b, orc_v3p spherical, orc_v3p disk, int nthreads);
/* a3 reflectDirection, a4 colorToLuminance */
void orc_batch_reflect_luminance(int64_t n, orc_cv3p i, orc_cv3p nrm, orc_cv3p color, orc_v3p reflected, float *luminance);

/* ---- synthetic input generator; bit-identical to the device
 *      generator in rlshaders_amd/csrc/gen.hip: only + - * / sqrt and integer hashing. ---- */
uint32_t orc_hash_u32(uint32_t seed, uint64_t index, uint32_t stream);
float    orc_hash_u01(uint32_t seed, uint64_t index, uint32_t stream);
/* stream ids */
enum {
    ORC_S_N0 = 0, ORC_S_N1, ORC_S_T, ORC_S_WO0, ORC_S_WO1,
    ORC_S_ROUGH, ORC_S_IOR, ORC_S_ANISO, ORC_S_KS_R, ORC_S_KS_G, ORC_S_KS_B,
    ORC_S_XI0, ORC_S_XI1, ORC_S_XI2, ORC_S_XI3, ORC_S_XI4, ORC_S_XI5,
    ORC_S_PARAM0 = 32,  /* closure specific scalars: PARAM0 + k */
    ORC_S_SCRAMBLE = 64 /* (0,2)-sequence scrambles: SCRAMBLE + 2*dim_pair + {0,1} */
};
void orc_gen_frame(uint32_t seed, uint64_t first, int64_t n, orc_v3p wo, orc_v3p N, orc_v3p T);
/* out[i] = lo + (hi-lo)*u01(seed, first+i, stream) */
void orc_gen_uniform(uint32_t seed, uint64_t first, int64_t n, uint32_t stream, float lo, float hi, float *out);
/* anisotropic rule of section 8(d): 0 for even index, U[0,1) for odd */
void orc_gen_aniso(uint32_t seed, uint64_t first, int64_t n, float *out);

int orc_hardware_threads(void);
/* src/rlGgx.h:152 (std::make_shared per closure): 1 = orc_ggx_init pays one 48-byte heap allocation + release, as the
 * reference's constructor does; values unchanged (bench.py's "port+alloc" CPU baseline leg).  Default 0. */
void orc_set_closure_alloc(int on);
int orc_get_closure_alloc(void);

#ifdef __cplusplus
}
#endif
#endif
