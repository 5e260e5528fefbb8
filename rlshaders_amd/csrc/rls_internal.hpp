// rls_internal.hpp -- host-side plumbing shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rlshaders_amd.h"

// RLS_DIAGNOSTICS (build option, default 1).  With 0 the library carries no measurement code at all: no `*_kernel_stamped`
// instantiation in any code object, no rls_diag_* symbol, no stamp test on a launch path -- and the product kernels are the
// same machine code either way, up to address literals (tests/test_diagnostics_option.py compares them).  bench.py's in-kernel clock needs 1.
#ifndef RLS_DIAGNOSTICS
#define RLS_DIAGNOSTICS 1
#endif
#if RLS_DIAGNOSTICS
#include "../../include/rlshaders_amd_diag.h"
#endif
#include "rls_device.hpp"

struct rls_context {
    int device;
    int compute_units;
    int blocks_per_cu;        // grid cap = compute_units * blocks_per_cu (RLS_BLOCKS_PER_CU)
    hipStream_t stream;       // stream launches go to
    hipStream_t own_stream;   // created by the context (may differ from `stream`)
    hipEvent_t ev_start, ev_stop;            // rls_timer_*
    hipEvent_t ev_probe_start, ev_probe_stop;   // placement probes (never the caller's timer)
    unsigned long long *scratch_u64;   // device, 8 bytes (checksum accumulator)
    int fast;                 // RLS_MATH_FAST selected (rls_context_set_math_mode)
    int capturing;            // between rls_graph_begin_capture and rls_graph_end_capture
    // rls_diag_clock_stamps_*: while `stamps` is set the four BASELINE kernels launch their stamped instantiation
    // (ClockStamp below); word 0 of the buffer holds the slot count, workgroup b writes words 4 + 4 b .. 7 + 4 b
    unsigned long long *stamps;       // = stamp_buf between rls_diag_clock_stamps_begin and _end, else NULL
    unsigned long long *stamp_buf;    // device, 8 * (4 + 4 * stamp_slots) bytes, allocated by the first _begin
    int64_t stamp_slots;
};

// Every kernel translation unit is compiled twice: with RLS_FAST=0 it carries the C ABI and the
// EXACT kernels, with RLS_FAST=1 only the FAST kernels behind one hidden dispatch symbol.
#define RLS_HIDDEN extern "C" __attribute__((visibility("hidden")))

namespace rlsh {

// ---- launch descriptors shared by the two builds of each unit ------------------------------------
enum GgxOp { OP_SAMPLE, OP_EVAL, OP_PDF, OP_FUSED, OP_REFRACT, OP_REFLECT_REFRACT, OP_MICROFACET, OP_NDF_PDF };
struct GgxIO {
    rls_ggx_closure c;
    const float *rx, *ry, *rx2, *ry2;
    rls_cvec3 cwi;
    rls_vec3 wi;
    rls_rgb f;
    float *pdf, *fresnel;
    rls_vec3 wt;
    float *weight;
    uint8_t *refracted;
    int64_t n;
    int kernel;
};

enum DisneyOp { DOP_SAMPLE, DOP_EVAL, DOP_PDF, DOP_FUSED };
struct DisneyIO {
    rls_disney_closure c;
    const float *rx, *ry;
    rls_cvec3 cwi;
    rls_vec3 wi;
    rls_rgb f;
    float *pdf;
    int64_t n;
};

enum SssOp { SOP_ND, SOP_ND_PDF, SOP_ND_EVAL, SOP_PROBE, SOP_MIS };
struct SssIO {
    rls_sss_closure c;
    const float *rx, *ry, *rin;
    rls_cvec3 P, disp, sampleN;
    int literal;
    float *r;
    rls_vec3 origin, dir;
    float *maxdist, *pdf;
    rls_rgb profile;
    int64_t n;
};
enum MiscOp { MOP_CAVITY, MOP_DIFFUSE_DIR, MOP_UTIL, MOP_REFLECT_LUM };
struct MiscIO {
    rls_cvec3 a, b, c;
    const float *rx, *ry;
    float *out;
    rls_vec3 v0, v1;
    int64_t n;
};

enum AltOp { AOP_GTR2_ANISO, AOP_GTR2, AOP_NDF_PDF, AOP_D_GTR2, AOP_GAUSS, AOP_LIBM };
struct AltIO {
    rls_disney_closure c;
    const float *rx, *ry;
    rls_cvec3 v;
    rls_vec3 out3;
    float *out1;
    rls_param dist_x;
    float *r, *pdf, *profile;
    int fn;                 // AOP_LIBM: RLS_FN_*
    int64_t n;
};

struct SkinIO {
    rls_skin_closure c;
    const float *xi[6];
    rls_skin_out o;
    int64_t n;
};

struct GgxIntIO {
    rls_ggx_closure c;
    rls_rgb sum;
    float *avgF;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;         // global index of point 0 (the sampler scrambles hash first + i)
};
struct DisneyIntIO {
    rls_disney_closure c;
    rls_rgb dsum, ssum;
    float *dcount, *scount;
    rls_disney_stream_out st;
    int streamed;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;         // global index of point 0 (the sampler scrambles hash first + i)
};

struct LightIO {
    rls_ggx_closure c;
    rls_ggx_shader sh;
    rls_cvec3 P;
    rls_sphere_light lights[RLS_MAX_LIGHTS];
    int nl;
    rls_rgb dd, ds;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;         // global index of point 0 (the sampler scrambles hash first + i)
};
struct GgxShadeIO {
    rls_ggx_closure c;
    rls_ggx_shader sh;
    rls_cvec3 P;
    rls_sphere_light lights[RLS_MAX_LIGHTS];
    int nl;
    float env[3];
    int traced;
    rls_rgb dd, ds, refr, id, is, out;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;
};
struct DisneyShadeIO {
    rls_disney_closure c;
    rls_cvec3 P;
    rls_sphere_light lights[RLS_MAX_LIGHTS];
    int nl;
    float env[3];
    rls_rgb dd, ds, id, is, out;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;
};
struct DisneyLightIO {
    rls_disney_closure c;
    rls_cvec3 P;
    rls_sphere_light lights[RLS_MAX_LIGHTS];
    int nl;
    rls_rgb dd, ds;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;
};

struct ScatterIO {
    rls_sss_closure c;
    rls_cvec3 P;
    rls_sss_scene scene;
    rls_rgb result;
    float *depth;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;         // global index of point 0 (the sampler scrambles hash first + i)
};

struct SkinIntIO {
    rls_skin_closure c;
    rls_cvec3 P;
    rls_sss_scene scene;
    float env[3];
    rls_sphere_light lights[RLS_MAX_LIGHTS];
    int nl;
    rls_rgb sheen, specular, sss, out;
    float *sheenFresnel, *specularFresnel, *sssWeight;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;
};
struct RefractIntIO {
    rls_ggx_closure c;
    float env[3];
    int traced;
    rls_rgb result;
    float *tir;
    int64_t n;
    int spp;
    uint32_t seed;
    uint64_t first;
};

void set_error(const char *fmt, ...);
rls_status hip_fail(hipError_t e, const char *what);

#define RLS_HIP_TRY(expr)                                                   \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return rlsh::hip_fail(_e, #expr);             \
    } while (0)

#define RLS_REQUIRE(cond, msg)                                              \
    do {                                                                    \
        if (!(cond)) {                                                      \
            rlsh::set_error("%s: %s", __func__, msg);                       \
            return RLS_ERR_INVALID_ARGUMENT;                                \
        }                                                                   \
    } while (0)

#ifndef RLS_BLOCK
#define RLS_BLOCK 256
#endif
constexpr int kBlock = RLS_BLOCK;   // 4 wavefronts of 64

// occupancy of the rlGgx kernels (waves per SIMD the register allocator must allow).  Left alone the reflect+refract
// kernel takes 72 VGPRs (7 waves) and keeps its 31 plane pointers alive by spilling scalar registers into vector
// lanes; at 6 waves (up to 80 VGPRs) it runs 1.8 % faster, at 4 or 8 slower (2.285 / 2.244 / 2.38 / 2.36 ms, one box)
// (round 3: re-measured per verb with the kernels as they are now -- ggx.hip, RLS_GGX_WAVES: eight for all but evalBrdf)
#ifndef RLS_WAVES_PER_EU
#define RLS_WAVES_PER_EU 6
#endif
#define RLS_KERNEL_ATTR __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_WAVES_PER_EU, RLS_WAVES_PER_EU)))

// Pointwise streaming launches: enough workgroups to fill 256 CUs several times over, capped so
// that very large batches grid-stride instead of queueing millions of workgroups.
inline hipError_t &pending_device_error()
{
    static thread_local hipError_t e = hipSuccess;
    return e;
}

// cap_mult: the rlGgx and rlSkin kernels run a tile per workgroup (no grid-stride loop) up to 16 x the context's cap --
// measured with RLS_BLOCKS_PER_CU = 64 against 1024 at 2^26 points (profiles/r03_blocks_per_cu.txt): evalPdf / evalBrdf alone
// -7.5 %, config 2 -1.0 %, rlSkin -1.3 %, reflect triple 0; the rlSss / NDProfile and rlDisney kernels LOSE 3-5 % (their
// per-workgroup staging of the libm tables is paid per tile then) and keep the cap.
#ifndef RLS_CAP_MULT
#define RLS_CAP_MULT 16          // tuning knob: 1 = every kernel under the context's cap
#endif
inline dim3 grid_for(const rls_context *ctx, int64_t n, int points_per_block = kBlock, int cap_mult = 1)
{
    // called right before every launch: make the context's device current for this host thread
    // (a host may drive several contexts, one per GPU, from one thread)
    // a failure (lost device) is kept for check_launch(), which every launch path calls right after the launch
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) pending_device_error() = e;
    int64_t want = (n + points_per_block - 1) / points_per_block;
    int64_t cap = (int64_t)ctx->compute_units * ctx->blocks_per_cu * cap_mult;
    if (want < 1) want = 1;
    if (want > cap) want = cap;
    // a multiple of the 8 XCDs whenever there is that much work: the pointwise kernels then give each XCD
    // one contiguous eighth of every plane (tile_range below)
    if (want >= 8) want = (want + 7) / 8 * 8;
    return dim3((unsigned)want);
}

// Launches of the kernels that hoist parameter-only arithmetic out of the tile loop (UNIFORM_*): the fewer threads, the more
// tiles share one evaluation of the hoisted part -- but the chip wants a few workgroups per CU in flight and some more
// queued behind them.  Measured at 2^16 .. 2^26 points on the four hoisting workloads (profiles/r03_uniform_sizes.txt): about
// eight tiles per thread, never fewer than 16 workgroups per CU (or than there are tiles), never more than the context's cap:
// 2^22 points run on 4 096 workgroups of 4 tiles, 2^24 on 8 192 of 8, 2^26 on the cap's 16 384 of 16.  The hoisting is
// worth 8-16 % from 2^20 points, its full 6-20 % from 2^24; below 2^20 (one tile per thread at most) such a launch costs what
// the per-point kernel costs.
#ifndef RLS_HOIST_TILES_PER_THREAD
#define RLS_HOIST_TILES_PER_THREAD 8
#endif
#ifndef RLS_HOIST_MIN_BLOCKS_PER_CU
#define RLS_HOIST_MIN_BLOCKS_PER_CU 16
#endif
inline dim3 grid_for_hoisting(const rls_context *ctx, int64_t n, int points_per_block = kBlock)
{
    static const int tpt = [] { const char *e = getenv("RLS_HOIST_TILES_PER_THREAD"); int v = e ? atoi(e) : 0; return v > 0 ? v : RLS_HOIST_TILES_PER_THREAD; }();
    static const int bpc = [] { const char *e = getenv("RLS_HOIST_MIN_BLOCKS_PER_CU"); int v = e ? atoi(e) : 0; return v > 0 ? v : RLS_HOIST_MIN_BLOCKS_PER_CU; }();
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) pending_device_error() = e;
    const int64_t tiles = (n + points_per_block - 1) / points_per_block;
    int64_t want = (tiles + tpt - 1) / tpt;
    const int64_t fill = (int64_t)ctx->compute_units * bpc, cap = (int64_t)ctx->compute_units * ctx->blocks_per_cu;
    if (want < fill) want = fill;
    if (want > cap) want = cap;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    if (want >= 8) want = (want + 7) / 8 * 8;
    return dim3((unsigned)want);
}

inline rls_status check_launch(const char *what)
{
    hipError_t &pend = pending_device_error();
    if (pend != hipSuccess) {                       // hipSetDevice failed before the launch (grid_for)
        hipError_t e0 = pend;
        pend = hipSuccess;
        (void)hipGetLastError();
        return hip_fail(e0, "hipSetDevice");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return RLS_OK;
}

inline bool has3(rls_cvec3 v) { return v.x && v.y && v.z; }
inline bool has3(rls_vec3 v) { return v.x && v.y && v.z; }
inline bool has3(rls_rgb v) { return v.r && v.g && v.b; }
inline bool none3(rls_cvec3 v) { return !v.x && !v.y && !v.z; }
inline bool ok_rgb(const rls_param_rgb &p) { return (p.r && p.g && p.b) || (!p.r && !p.g && !p.b); }
inline bool ok_materials(const rls_material_index &m) { return m.id == nullptr || m.count > 0; }

// Host side of the stamps: the buffer a launch hands to a `*_kernel_stamped` instantiation, or NULL when the product kernel is
// to run (rls_diag_clock_stamps_begin not in force).  The slots are cleared on the launch stream first, so _read returns the
// stamps of the LAST stamped launch only, whatever grid an earlier stamped launch of the same begin/end bracket used.  A
// bracket and a graph recording exclude each other (context.hip), so a stamped launch is never baked into a graph.
// (Should the clear fail, the product kernel runs and _read reports no stamps: a diagnostic never costs the caller its launch.)
#if RLS_DIAGNOSTICS
inline unsigned long long *stamps_for_launch(rls_context *ctx)
{
    if (!ctx->stamps || ctx->capturing) return nullptr;
    if (hipMemsetAsync(ctx->stamps + 4, 0, sizeof(unsigned long long) * 4 * (size_t)ctx->stamp_slots, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return ctx->stamps;
}
#endif

} // namespace rlsh

// device-side views of the ABI structs ---------------------------------------------------------
namespace rlsd {

// The tiles (kBlock consecutive points) a workgroup of a pointwise kernel walks.  Workgroups are dealt
// round-robin to the 8 XCDs, each with its own L2 and its own path to HBM: with the plain grid-stride mapping
// every XCD touches every eighth kilobyte of every plane; here XCD x owns the x-th contiguous eighth and its
// workgroups stride through that.  Measured on the arithmetic-free 19-in / 12-out pattern: 1.64 -> 1.58 ms
// (tools/micro/streams31.hip).  Falls back to the plain mapping when the grid is not a multiple of 8.
struct TileRange { int64_t first, end, step; };
RLS_DEV TileRange tile_range(int64_t n)
{
    TileRange t;
    if (gridDim.x % 8u == 0u) {
        const int64_t tiles = (n + rlsh::kBlock - 1) / rlsh::kBlock;
        const int64_t per_xcd = (tiles + 7) / 8;
        const int64_t x = blockIdx.x % 8u, b = blockIdx.x / 8u;
        t.first = (x * per_xcd + b) * rlsh::kBlock;
        t.end = (x + 1) * per_xcd * rlsh::kBlock;
        if (t.end > n) t.end = n;
        t.step = (int64_t)(gridDim.x / 8u) * rlsh::kBlock;
    } else {
        t.first = (int64_t)blockIdx.x * rlsh::kBlock;
        t.end = n;
        t.step = (int64_t)gridDim.x * rlsh::kBlock;
    }
    return t;
}

// A fresh copy of the kernel's argument struct, read from the kernarg segment at the point of the call.  The pointwise
// kernels stream 25-59 planes: kept alive across the tile loop, their base pointers alone need up to 118 of the 102
// scalar registers, and the allocator spills them into vector lanes (v_writelane / v_readlane -- VALU instructions, on
// kernels that are bound by VALU issue: rlSkin 246 spilled SGPRs, reflect+refract 52, the rlSss probe 30).  Re-read per
// tile -- the input planes' pointers where the loads are issued, the output planes' where the stores are -- they live
// for a few instructions each; the s_load_dwordx4/x8 that fetch them go to the scalar cache and cost no vector issue.
// The empty asm makes the pointer opaque, so the loads cannot be hoisted back out of the loop.  IO must be the kernel's
// first (only) parameter.  Measured (profiles/r03_reload_args.txt, two libraries interleaved on one box): rlSkin 246 -> 42
// spilled SGPRs, 4.44 -> 4.34 ms (-2.2 %); the rlSss probe 30 -> 0, 1.710 -> 1.686 ms (-1.4 %); the rlGgx kernels 52 -> 2:
// +2.3 % when first measured at six waves per SIMD (2.054 -> 2.102 ms), but since those kernels run at eight waves (64
// vector registers, where every spilled scalar costs a lane write) the reload is worth 7 %: 2.139 ms without it, 2.001 ms with
// (profiles/r04_ggx_ab.txt, tools/ab.sh, two interleaved repetitions) -- every pointwise unit uses it.
template <class IO>
RLS_DEV IO reload_args(const IO &a)
{
#if defined(__HIP_DEVICE_COMPILE__)     // (the host pass of the translation unit only parses this)
    typedef const __attribute__((address_space(4))) IO *KP;
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
#else
    return a;
#endif
}

// In-kernel clock stamps, for the DIAGNOSTIC instantiations of the four BASELINE kernels only (`*_kernel_stamped`, launched
// while rls_diag_clock_stamps_begin is in force; the product kernels share the body and contain no counter read -- the
// disassembly before and after the split: profiles/r05_stamped_isa.txt).  Wave 0 of
// every workgroup reads the shader-clock counter (s_memtime: one tick per shader cycle) and the constant 100 MHz counter
// (s_memrealtime) on entry and on exit; effective clock of that workgroup's lifetime = d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  The stamps go to a buffer of their own; no output depends on them.
#if RLS_DIAGNOSTICS
template <int STAMP> struct ClockStamp {
    RLS_DEV void begin() {}
    RLS_DEV void end(unsigned long long *) {}
};
template <> struct ClockStamp<1> {
    unsigned long long t0, r0;
    RLS_DEV void begin()
    {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    RLS_DEV void end(unsigned long long *buf)
    {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && buf != nullptr && (unsigned long long)blockIdx.x < buf[0]) {
            unsigned long long *w = buf + 4 + 4 * (size_t)blockIdx.x;
            w[0] = t0; w[1] = t1; w[2] = r0; w[3] = r1;
        }
    }
};
#endif

RLS_DEV int64_t idx_full(int64_t i) { return i; }
RLS_DEV int64_t idx_full(const Idx &i) { return i.full(); }
// I: int64_t or Idx.  STREAMED: every optional parameter plane is present (checked on the host), so the
// per-parameter "stream or uniform" test -- a scalar branch per parameter per iteration -- disappears.
template <bool STREAMED = false, class I>
RLS_DEV float ldp(const rls_param &p, I i) { return (STREAMED || p.v) ? ldg(p.v, i) : p.u; }
// Parameters by reference (rls_material_index): the index a PARAMETER load uses -- the point itself (per-point planes) or the
// point's material id (per-material columns; clamped, so a hostile id reads the last entry instead of past the table).
// INDEXED = false compiles the lookup away (kernels that are only launched without a table).
template <class I>
struct PIndex { I i; bool indexed; uint32_t id; };
template <bool INDEXED = true, class I>
RLS_DEV PIndex<I> pindex(const rls_material_index &m, I i)
{
    PIndex<I> k = { i, false, 0u };
    if (INDEXED && m.id != nullptr) {
        k.indexed = true;
        const uint32_t id = m.id[idx_full(i)];
        k.id = id < m.count ? id : m.count - 1u;
    }
    return k;
}
template <bool STREAMED = false, class I>
RLS_DEV float ldp(const rls_param &p, const PIndex<I> &k)
{
    if (STREAMED) return ldg(p.v, k.i);
    if (!p.v) return p.u;
    return k.indexed ? p.v[k.id] : ldg(p.v, k.i);
}
template <bool STREAMED = false, class I>
RLS_DEV void ldrgb(const rls_param_rgb &p, const PIndex<I> &k, float &r, float &g, float &b)
{
    if (STREAMED) { r = ldg(p.r, k.i); g = ldg(p.g, k.i); b = ldg(p.b, k.i); }
    else if (!p.r) { r = p.ur; g = p.ug; b = p.ub; }
    else if (k.indexed) { r = p.r[k.id]; g = p.g[k.id]; b = p.b[k.id]; }
    else { r = ldg(p.r, k.i); g = ldg(p.g, k.i); b = ldg(p.b, k.i); }
}
template <class I>
RLS_DEV V3 ld3(const rls_cvec3 &p, I i) { return mk(ldg(p.x, i), ldg(p.y, i), ldg(p.z, i)); }
template <class I>
RLS_DEV void st3(const rls_vec3 &p, I i, V3 v) { stg(p.x, i, v.x); stg(p.y, i, v.y); stg(p.z, i, v.z); }
template <class I>
RLS_DEV void strgb(const rls_rgb &p, I i, float r, float g, float b)
{
    stg(p.r, i, r); stg(p.g, i, g); stg(p.b, i, b);
}
template <bool STREAMED = false, class I>
RLS_DEV void ldrgb(const rls_param_rgb &p, I i, float &r, float &g, float &b)
{
    if (STREAMED || p.r) { r = ldg(p.r, i); g = ldg(p.g, i); b = ldg(p.b, i); }
    else { r = p.ur; g = p.ug; b = p.ub; }
}

} // namespace rlsd
