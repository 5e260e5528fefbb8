// rls_internal.hpp -- host-side plumbing shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rlshaders_amd.h"
#include "rls_device.hpp"

struct rls_context {
    int device;
    int compute_units;
    int blocks_per_cu;        // grid cap = compute_units * blocks_per_cu (RLS_BLOCKS_PER_CU)
    hipStream_t stream;       // stream launches go to
    hipStream_t own_stream;   // created by the context (may differ from `stream`)
    hipEvent_t ev_start, ev_stop;
    unsigned long long *scratch_u64;   // device, 8 bytes (checksum accumulator)
};

namespace rlsh {

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

// occupancy hint for the closure kernels (waves per SIMD the register allocator must allow)
#ifdef RLS_WAVES_PER_EU
#define RLS_KERNEL_ATTR __launch_bounds__(rlsh::kBlock) __attribute__((amdgpu_waves_per_eu(RLS_WAVES_PER_EU, RLS_WAVES_PER_EU)))
#else
#define RLS_KERNEL_ATTR __launch_bounds__(rlsh::kBlock)
#endif

// Pointwise streaming launches: enough workgroups to fill 256 CUs several times over, capped so
// that very large batches grid-stride instead of queueing millions of workgroups.
inline dim3 grid_for(const rls_context *ctx, int64_t n, int points_per_block = kBlock)
{
    int64_t want = (n + points_per_block - 1) / points_per_block;
    int64_t cap = (int64_t)ctx->compute_units * ctx->blocks_per_cu;
    if (want < 1) want = 1;
    return dim3((unsigned)(want < cap ? want : cap));
}

inline rls_status check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return RLS_OK;
}

inline bool has3(rls_cvec3 v) { return v.x && v.y && v.z; }
inline bool has3(rls_vec3 v) { return v.x && v.y && v.z; }
inline bool has3(rls_rgb v) { return v.r && v.g && v.b; }
inline bool none3(rls_cvec3 v) { return !v.x && !v.y && !v.z; }
inline bool ok_rgb(const rls_param_rgb &p) { return (p.r && p.g && p.b) || (!p.r && !p.g && !p.b); }

} // namespace rlsh

// device-side views of the ABI structs ---------------------------------------------------------
namespace rlsd {

RLS_DEV float ldp(const rls_param &p, int64_t i) { return p.v ? ldg(p.v, i) : p.u; }
RLS_DEV V3 ld3(const rls_cvec3 &p, int64_t i) { return mk(ldg(p.x, i), ldg(p.y, i), ldg(p.z, i)); }
RLS_DEV void st3(const rls_vec3 &p, int64_t i, V3 v) { stg(p.x, i, v.x); stg(p.y, i, v.y); stg(p.z, i, v.z); }
RLS_DEV void strgb(const rls_rgb &p, int64_t i, float r, float g, float b)
{
    stg(p.r, i, r); stg(p.g, i, g); stg(p.b, i, b);
}
RLS_DEV void ldrgb(const rls_param_rgb &p, int64_t i, float &r, float &g, float &b)
{
    if (p.r) { r = ldg(p.r, i); g = ldg(p.g, i); b = ldg(p.b, i); }
    else { r = p.ur; g = p.ug; b = p.ub; }
}

} // namespace rlsd
