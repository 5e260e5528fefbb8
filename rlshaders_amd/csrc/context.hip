// context.hip -- context, memory, timing, synthetic generator and checksum entry points of the
// C ABI (include/rlshaders_amd.h).  gfx950 only.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "rls_internal.hpp"

namespace {
thread_local char g_error[512] = "";
}

namespace rlsh {

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

rls_status hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    // the failure is reported through our status; do not leave it as the runtime's "last error" for
    // whoever shares the process (a framework would otherwise see e.g. a stale out-of-memory)
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? RLS_ERR_OUT_OF_MEMORY : RLS_ERR_HIP;
}

} // namespace rlsh

using namespace rlsd;

extern "C" {

const char *rls_last_error(void) { return g_error; }

const char *rls_status_string(rls_status s)
{
    switch (s) {
    case RLS_OK: return "ok";
    case RLS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RLS_ERR_NO_DEVICE: return "no HIP device";
    case RLS_ERR_HIP: return "HIP runtime error";
    case RLS_ERR_OUT_OF_MEMORY: return "out of device memory";
    case RLS_ERR_UNSUPPORTED: return "unsupported";
    case RLS_ERR_ABORTED: return "stopped by a callback";
    default: return "unknown status";
    }
}

int rls_version(void) { return RLS_VERSION_MAJOR * 1000 + RLS_VERSION_MINOR; }

rls_status rls_context_create(int device_ordinal, rls_context **out)
{
    RLS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        rlsh::set_error("rls_context_create: no HIP device visible (%s)",
                        e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return RLS_ERR_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= count) {
        rlsh::set_error("rls_context_create: device %d out of range [0, %d)", device_ordinal, count);
        return RLS_ERR_NO_DEVICE;
    }
    RLS_HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    RLS_HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));

    rls_context *ctx = (rls_context *)calloc(1, sizeof(rls_context));
    if (!ctx) { rlsh::set_error("rls_context_create: host allocation failed"); return RLS_ERR_OUT_OF_MEMORY; }
    ctx->device = device_ordinal;
    ctx->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->blocks_per_cu = 64;   // measured on MI355X: 8 -> 3.01 ms, 16 -> 2.85, 64 -> 2.72, 1024 -> 2.70 (config 2)
    if (const char *s = getenv("RLS_BLOCKS_PER_CU")) {
        int v = atoi(s);
        if (v >= 1 && v <= 4096) ctx->blocks_per_cu = v;
    }
    hipError_t e1 = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    hipError_t e2 = e1 == hipSuccess ? hipEventCreate(&ctx->ev_start) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipEventCreate(&ctx->ev_stop) : e2;
    if (e3 == hipSuccess) e3 = hipEventCreate(&ctx->ev_probe_start);
    if (e3 == hipSuccess) e3 = hipEventCreate(&ctx->ev_probe_stop);
    hipError_t e4 = e3 == hipSuccess ? hipMalloc((void **)&ctx->scratch_u64, sizeof(unsigned long long)) : e3;
    if (e4 != hipSuccess) {
        rls_status st = rlsh::hip_fail(e4, "rls_context_create");
        rls_context_destroy(ctx);
        return st;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return RLS_OK;
}

void rls_context_destroy(rls_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // work queued on the context's own stream must not outlive the context's scratch memory; a
    // caller-provided stream is the caller's to drain
    if (ctx->capturing) {   // abandon an unfinished recording
        hipGraph_t g = nullptr;
        if (hipStreamEndCapture(ctx->stream, &g) == hipSuccess && g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
    }
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->scratch_u64) (void)hipFree(ctx->scratch_u64);
    if (ctx->stamp_buf) (void)hipFree(ctx->stamp_buf);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_probe_start) (void)hipEventDestroy(ctx->ev_probe_start);
    if (ctx->ev_probe_stop) (void)hipEventDestroy(ctx->ev_probe_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    free(ctx);
}

int rls_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

rls_status rls_shard_range(int64_t total, int rank, int world, int64_t *first, int64_t *count)
{
    RLS_REQUIRE(first != nullptr && count != nullptr, "NULL argument");
    RLS_REQUIRE(world >= 1 && rank >= 0 && rank < world && total >= 0, "bad shard request");
    // 128-bit products: total * (rank + 1) can exceed 63 bits for absurd totals only, but cost nothing to get right
    const __int128 lo = (__int128)total * rank / world, hi = (__int128)total * (rank + 1) / world;
    *first = (int64_t)lo;
    *count = (int64_t)(hi - lo);
    return RLS_OK;
}

rls_status rls_context_set_stream(rls_context *ctx, void *hip_stream)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    ctx->stream = (hipStream_t)hip_stream;
    return RLS_OK;
}

rls_status rls_context_use_own_stream(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    ctx->stream = ctx->own_stream;
    return RLS_OK;
}

void *rls_context_get_stream(rls_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

rls_status rls_context_set_math_mode(rls_context *ctx, int mode)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(mode == RLS_MATH_EXACT || mode == RLS_MATH_FAST, "mode must be RLS_MATH_EXACT or RLS_MATH_FAST");
    ctx->fast = mode == RLS_MATH_FAST;
    return RLS_OK;
}

int rls_context_get_math_mode(const rls_context *ctx) { return ctx && ctx->fast ? RLS_MATH_FAST : RLS_MATH_EXACT; }

int rls_context_device(const rls_context *ctx) { return ctx ? ctx->device : -1; }

rls_status rls_context_synchronize(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RLS_OK;
}

rls_status rls_device_info(rls_context *ctx, int *compute_units, size_t *hbm_total, size_t *hbm_free,
                           char *arch_name, size_t arch_name_len)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    if (compute_units) *compute_units = ctx->compute_units;
    if (hbm_total || hbm_free) {
        size_t f = 0, t = 0;
        RLS_HIP_TRY(hipMemGetInfo(&f, &t));
        if (hbm_total) *hbm_total = t;
        if (hbm_free) *hbm_free = f;
    }
    if (arch_name && arch_name_len > 0) {
        hipDeviceProp_t prop;
        RLS_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
        strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = '\0';
    }
    return RLS_OK;
}

rls_status rls_device_alloc(rls_context *ctx, size_t bytes, void **out)
{
    RLS_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    *out = nullptr;
    if (bytes == 0) return RLS_OK;
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipMalloc(out, bytes));
    return RLS_OK;
}

rls_status rls_device_free(rls_context *ctx, void *p)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    if (!p) return RLS_OK;
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipFree(p));
    return RLS_OK;
}

rls_status rls_copy_to_device(rls_context *ctx, void *dst, const void *src_host, size_t bytes)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    if (bytes == 0) return RLS_OK;
    RLS_REQUIRE(dst != nullptr && src_host != nullptr, "NULL buffer");
    RLS_HIP_TRY(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    RLS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RLS_OK;
}

rls_status rls_copy_to_host(rls_context *ctx, void *dst_host, const void *src, size_t bytes)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    if (bytes == 0) return RLS_OK;
    RLS_REQUIRE(dst_host != nullptr && src != nullptr, "NULL buffer");
    RLS_HIP_TRY(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RLS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RLS_OK;
}

rls_status rls_timer_start(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    return RLS_OK;
}

rls_status rls_timer_stop(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    return RLS_OK;
}

rls_status rls_timer_elapsed_ms(rls_context *ctx, float *ms)
{
    RLS_REQUIRE(ctx != nullptr && ms != nullptr, "NULL argument");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_HIP_TRY(hipEventSynchronize(ctx->ev_stop));
    RLS_HIP_TRY(hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return RLS_OK;
}

#if RLS_DIAGNOSTICS
// ---- in-kernel clock stamps (diagnostic) -----------------------------------------------------
// rls_internal.hpp, ClockStamp: while in force, rls_ggx_reflect_refract (all planes streamed), rls_sss_probe_ray
// (per-point distances), rls_skin_sample_eval_pdf (all planes streamed) and rls_disney_integrate (one lane per point)
// launch the stamped instantiation of their kernel; every other entry point launches what it always launches.
rls_status rls_diag_clock_stamps_begin(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->stamp_buf) {
        // one slot per workgroup of the largest grid any pointwise launch uses (grid_for: the cap x RLS_CAP_MULT, rounded
        // up to a multiple of 8)
        ctx->stamp_slots = (int64_t)ctx->compute_units * ctx->blocks_per_cu * RLS_CAP_MULT + 8;
        RLS_HIP_TRY(hipMalloc((void **)&ctx->stamp_buf, sizeof(unsigned long long) * (size_t)(4 + 4 * ctx->stamp_slots)));
    }
    RLS_HIP_TRY(hipMemsetAsync(ctx->stamp_buf, 0, sizeof(unsigned long long) * (size_t)(4 + 4 * ctx->stamp_slots), ctx->stream));
    const unsigned long long slots = (unsigned long long)ctx->stamp_slots;
    RLS_HIP_TRY(hipMemcpyAsync(ctx->stamp_buf, &slots, sizeof(slots), hipMemcpyHostToDevice, ctx->stream));
    RLS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stamps = ctx->stamp_buf;
    return RLS_OK;
}

rls_status rls_diag_clock_stamps_read(rls_context *ctx, int64_t capacity, uint64_t *stamps_host, int64_t *count)
{
    RLS_REQUIRE(ctx != nullptr && count != nullptr, "NULL argument");
    RLS_REQUIRE(ctx->stamps != nullptr, "rls_diag_clock_stamps_begin is not in force");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_REQUIRE(capacity >= 0 && (capacity == 0 || stamps_host != nullptr), "capacity < 0 or stamps_host is NULL");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "stamp words are 64 bits");
    const int64_t take = capacity < ctx->stamp_slots ? capacity : ctx->stamp_slots;
    *count = ctx->stamp_slots;       // slots there are; workgroups that never ran leave theirs zero
    if (take > 0) {
        RLS_HIP_TRY(hipMemcpyAsync(stamps_host, ctx->stamp_buf + 4, sizeof(uint64_t) * 4 * (size_t)take, hipMemcpyDeviceToHost,
                                   ctx->stream));
        RLS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return RLS_OK;
}

rls_status rls_diag_clock_stamps_end(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->stamps = nullptr;
    return RLS_OK;
}
#endif

// ---- launch graphs ---------------------------------------------------------------------------
struct rls_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    int device;
};

rls_status rls_graph_begin_capture(rls_context *ctx)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(ctx->stream != nullptr, "the NULL stream cannot be captured: use the context's own stream");
    RLS_REQUIRE(!ctx->capturing, "a capture is already in progress on this context");
    // a recorded stamped launch would keep writing stamps on every replay, long after rls_diag_clock_stamps_end
    RLS_REQUIRE(ctx->stamps == nullptr, "not allowed between rls_diag_clock_stamps_begin and _end");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    // thread-local: other host threads driving other contexts keep running normally
    RLS_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = 1;
    return RLS_OK;
}

rls_status rls_graph_end_capture(rls_context *ctx, rls_graph **out)
{
    RLS_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    RLS_REQUIRE(ctx->capturing, "no capture in progress");
    *out = nullptr;
    ctx->capturing = 0;
    hipGraph_t graph = nullptr;
    RLS_HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return rlsh::hip_fail(e, "hipGraphInstantiate");
    }
    rls_graph *g = (rls_graph *)calloc(1, sizeof(rls_graph));
    if (!g) {
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        rlsh::set_error("rls_graph_end_capture: out of host memory");
        return RLS_ERR_HIP;
    }
    g->graph = graph; g->exec = exec; g->device = ctx->device;
    *out = g;
    return RLS_OK;
}

rls_status rls_graph_launch(rls_context *ctx, rls_graph *graph)
{
    RLS_REQUIRE(ctx != nullptr && graph != nullptr, "NULL argument");
    RLS_REQUIRE(graph->device == ctx->device, "graph was recorded on another device");
    RLS_REQUIRE(!ctx->capturing, "a capture is in progress on this context");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipGraphLaunch(graph->exec, ctx->stream));
    return RLS_OK;
}

void rls_graph_destroy(rls_graph *graph)
{
    if (!graph) return;
    (void)hipGraphExecDestroy(graph->exec);
    (void)hipGraphDestroy(graph->graph);
    free(graph);
}

} // extern "C"

// ---- generator kernels -----------------------------------------------------------------------
namespace {

// hash stream ids (DESIGN.md "Synthetic inputs")
enum { S_N0 = 0, S_N1, S_T, S_WO0, S_WO1, S_ROUGH, S_IOR, S_ANISO };

__device__ __forceinline__ void circle_point(float u, float &c, float &s)
{
    float t = 4.0f * u;
    int q = (int)t;
    float f = t - (float)q;
    float a = 1.0f - f, b = f;
    float l = sqrtf(a * a + b * b);
    a = a / l; b = b / l;
    switch (q & 3) {
    case 0: c = a;  s = b;  break;
    case 1: c = -b; s = a;  break;
    case 2: c = -a; s = -b; break;
    default: c = b; s = -a; break;
    }
}

__device__ __forceinline__ V3 normalize_div(V3 a)
{
    float l = length(a);
    return mk(a.x / l, a.y / l, a.z / l);
}

__global__ __launch_bounds__(rlsh::kBlock) void gen_frame_kernel(uint32_t seed, uint64_t first, int64_t n,
                                                                 rls_vec3 wo, rls_vec3 N, rls_vec3 T)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        uint64_t i = first + (uint64_t)k;
        float u0 = hash_u01(seed, i, S_N0);
        float z = 1.0f - 2.0f * u0;
        float rr = sqrtf(maxf(0.0f, 1.0f - z * z));
        float c, s;
        circle_point(hash_u01(seed, i, S_N1), c, s);
        V3 Nn = normalize_div(mk(rr * c, rr * s, z));

        V3 e = absf(Nn.x) < 0.57735f ? mk(1.0f, 0.0f, 0.0f) : mk(0.0f, 1.0f, 0.0f);
        V3 T0 = normalize_div(cross(e, Nn));
        V3 B0 = cross(Nn, T0);
        circle_point(hash_u01(seed, i, S_T), c, s);
        V3 Tt = T0 * c + B0 * s;
        Tt = Tt - Nn * dot(Tt, Nn);
        Tt = normalize_div(Tt);
        V3 Bt = cross(Nn, Tt);

        float ct = 0.02f + 0.98f * hash_u01(seed, i, S_WO0);
        float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
        circle_point(hash_u01(seed, i, S_WO1), c, s);
        V3 w = (Tt * c + Bt * s) * st + Nn * ct;
        w = normalize_div(w);

        N.x[k] = Nn.x; N.y[k] = Nn.y; N.z[k] = Nn.z;
        T.x[k] = Tt.x; T.y[k] = Tt.y; T.z[k] = Tt.z;
        wo.x[k] = w.x; wo.y[k] = w.y; wo.z[k] = w.z;
    }
}

__global__ __launch_bounds__(rlsh::kBlock) void gen_uniform_kernel(uint32_t seed, uint64_t first, int64_t n,
                                                                   uint32_t stream, float lo, float span, float *out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        out[k] = lo + span * hash_u01(seed, first + (uint64_t)k, stream);
    }
}

__global__ __launch_bounds__(rlsh::kBlock) void gen_aniso_kernel(uint32_t seed, uint64_t first, int64_t n, float *out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        uint64_t i = first + (uint64_t)k;
        out[k] = (i & 1ULL) ? hash_u01(seed, i, S_ANISO) : 0.0f;
    }
}

// Order-independent checksum: sum over elements of a 64-bit hash of (bit pattern, low index bits
// are NOT mixed in, so the same multiset of values gives the same sum in any order).
__global__ __launch_bounds__(rlsh::kBlock) void checksum_kernel(int64_t n, const float *data, unsigned long long *acc)
{
    unsigned long long local = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        uint32_t b = __float_as_uint(data[k]);
        uint64_t h = ((uint64_t)mix32(b ^ 0x68bc21ebU) << 32) | (uint64_t)mix32(b + 0x02e5be93U);
        local += h;
    }
    // wave64 butterfly, then one atomic per wavefront
    for (int off = 32; off > 0; off >>= 1) {
        local += __shfl_xor(local, off, 64);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, local);
}

// Placement probe: the access pattern of the closure kernels -- 19 planes read, 12 written, one float per
// lane per plane, XCD-contiguous tiles -- with no arithmetic, over a block treated as 31 equal sub-planes.
struct ProbePlanes { const float *in[19]; float *out[12]; };
__global__ __launch_bounds__(rlsh::kBlock) void placement_probe_kernel(ProbePlanes p, int64_t n)
{
    const rlsd::TileRange tiles = rlsd::tile_range(n);
    for (int64_t base = tiles.first; base < tiles.end; base += tiles.step) {
        const int64_t i = base + threadIdx.x;
        if (i >= n) continue;
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < 19; j++) a += __builtin_nontemporal_load(p.in[j] + i);
#pragma unroll
        for (int j = 0; j < 12; j++) __builtin_nontemporal_store(a + (float)j, p.out[j] + i);
    }
}

rls_status probe_block(rls_context *ctx, void *block, size_t bytes, float *gbs)
{
    const int64_t n = (int64_t)(bytes / (31 * sizeof(float))) / 64 * 64;   // sub-planes on 256-byte boundaries
    *gbs = 0.0f;
    if (n < rlsh::kBlock) return RLS_OK;
    ProbePlanes p;
    float *base = (float *)block;
    for (int j = 0; j < 19; j++) p.in[j] = base + (int64_t)j * n;
    for (int j = 0; j < 12; j++) p.out[j] = base + (int64_t)(19 + j) * n;
    const dim3 grid = rlsh::grid_for(ctx, n);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(placement_probe_kernel, grid, dim3(rlsh::kBlock), 0, ctx->stream, p, n);
    // the probe's own event pair: rls_timer_start ... rls_timer_elapsed_ms of the caller stays intact across a probe
    RLS_HIP_TRY(hipEventRecord(ctx->ev_probe_start, ctx->stream));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(placement_probe_kernel, grid, dim3(rlsh::kBlock), 0, ctx->stream, p, n);
    RLS_HIP_TRY(hipEventRecord(ctx->ev_probe_stop, ctx->stream));
    rls_status st = rlsh::check_launch("placement_probe_kernel");
    if (st != RLS_OK) return st;
    RLS_HIP_TRY(hipEventSynchronize(ctx->ev_probe_stop));
    float ms = 0.0f;
    RLS_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_probe_start, ctx->ev_probe_stop));
    if (ms > 0.0f) *gbs = (float)(31.0 * sizeof(float) * (double)n * reps / (ms * 1e-3) / 1e9);
    return RLS_OK;
}

} // namespace

struct rls_arena {
    int device;
    void *block;
    size_t bytes, plane_bytes;
    int planes, candidates;
    float probe_gbs, probe_gbs_min, probe_gbs_max;
};

extern "C" {

rls_status rls_probe_block(rls_context *ctx, void *block, size_t bytes, float *gb_per_s)
{
    RLS_REQUIRE(ctx != nullptr && block != nullptr && gb_per_s != nullptr, "NULL argument");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipMemsetAsync(block, 0, bytes, ctx->stream));
    return probe_block(ctx, block, bytes, gb_per_s);
}

rls_status rls_arena_create(rls_context *ctx, int64_t n, int planes, int candidates, rls_arena **out)
{
    RLS_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_REQUIRE(n > 0 && planes > 0, "n and planes must be positive");
    RLS_REQUIRE(candidates >= 1 && candidates <= 16, "candidates must be in [1, 16]");
    *out = nullptr;
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    // planes start on 256-byte boundaries
    const size_t plane_bytes = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    const size_t bytes = plane_bytes * (size_t)planes;
    void *blocks[16] = {nullptr};
    float gbs[16] = {0.0f};
    int got = 0;
    rls_status st = RLS_OK;
    for (int k = 0; k < candidates; k++) {
        hipError_t e = hipMalloc(&blocks[k], bytes);
        if (e != hipSuccess) {                      // fewer candidates than asked for is fine, none is not
            (void)hipGetLastError();
            if (k == 0) st = rlsh::hip_fail(e, "hipMalloc(arena)");
            break;
        }
        got++;
        hipError_t m = hipMemsetAsync(blocks[k], 0, bytes, ctx->stream);
        if (m != hipSuccess) { st = rlsh::hip_fail(m, "hipMemsetAsync(arena)"); break; }
        if (candidates > 1) {
            st = probe_block(ctx, blocks[k], bytes, &gbs[k]);
            if (st != RLS_OK) break;
        }
    }
    int best = 0;
    float lo = gbs[0], hi = gbs[0];
    for (int k = 1; k < got; k++) {
        if (gbs[k] > gbs[best]) best = k;
        lo = gbs[k] < lo ? gbs[k] : lo;
        hi = gbs[k] > hi ? gbs[k] : hi;
    }
    rls_arena *a = (st == RLS_OK && got > 0) ? (rls_arena *)calloc(1, sizeof(rls_arena)) : nullptr;
    for (int k = 0; k < got; k++) {
        if (!a || k != best) (void)hipFree(blocks[k]);
    }
    if (!a) {
        if (st == RLS_OK) { rlsh::set_error("rls_arena_create: out of host memory"); st = RLS_ERR_OUT_OF_MEMORY; }
        return st;
    }
    hipError_t fin = hipSuccess;
    if (candidates > 1) fin = hipMemsetAsync(blocks[best], 0, bytes, ctx->stream);          // the probe wrote into it
    if (fin == hipSuccess) fin = hipStreamSynchronize(ctx->stream);
    if (fin != hipSuccess) {                                                                // do not leak the block
        (void)hipFree(blocks[best]);
        free(a);
        return rlsh::hip_fail(fin, "rls_arena_create: clearing the chosen block");
    }
    a->device = ctx->device; a->block = blocks[best]; a->bytes = bytes; a->plane_bytes = plane_bytes;
    a->planes = planes; a->candidates = got;
    a->probe_gbs = gbs[best]; a->probe_gbs_min = lo; a->probe_gbs_max = hi;
    *out = a;
    return RLS_OK;
}

float *rls_arena_plane(const rls_arena *arena, int k)
{
    if (!arena || k < 0 || k >= arena->planes) return nullptr;
    return (float *)((char *)arena->block + (size_t)k * arena->plane_bytes);
}

rls_status rls_arena_info(const rls_arena *arena, size_t *bytes, int *candidates_probed, float *probe_gb_per_s,
                          float *probe_min, float *probe_max)
{
    RLS_REQUIRE(arena != nullptr, "arena is NULL");
    if (bytes) *bytes = arena->bytes;
    if (candidates_probed) *candidates_probed = arena->candidates;
    if (probe_gb_per_s) *probe_gb_per_s = arena->probe_gbs;
    if (probe_min) *probe_min = arena->probe_gbs_min;
    if (probe_max) *probe_max = arena->probe_gbs_max;
    return RLS_OK;
}

void rls_arena_destroy(rls_arena *arena)
{
    if (!arena) return;
    (void)hipSetDevice(arena->device);
    (void)hipFree(arena->block);
    free(arena);
}

rls_status rls_gen_frame(rls_context *ctx, uint32_t seed, uint64_t first_index, int64_t n,
                         rls_vec3 wo, rls_vec3 N, rls_vec3 T)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(rlsh::has3(wo) && rlsh::has3(N) && rlsh::has3(T), "NULL output plane");
    hipLaunchKernelGGL(gen_frame_kernel, rlsh::grid_for(ctx, n), dim3(rlsh::kBlock), 0, ctx->stream,
                       seed, first_index, n, wo, N, T);
    return rlsh::check_launch("gen_frame_kernel");
}

rls_status rls_gen_uniform(rls_context *ctx, uint32_t seed, uint64_t first_index, int64_t n,
                           uint32_t stream, float lo, float hi, float *out)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(out != nullptr, "out is NULL");
    hipLaunchKernelGGL(gen_uniform_kernel, rlsh::grid_for(ctx, n), dim3(rlsh::kBlock), 0, ctx->stream,
                       seed, first_index, n, stream, lo, hi - lo, out);
    return rlsh::check_launch("gen_uniform_kernel");
}

rls_status rls_gen_aniso(rls_context *ctx, uint32_t seed, uint64_t first_index, int64_t n, float *out)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(out != nullptr, "out is NULL");
    hipLaunchKernelGGL(gen_aniso_kernel, rlsh::grid_for(ctx, n), dim3(rlsh::kBlock), 0, ctx->stream,
                       seed, first_index, n, out);
    return rlsh::check_launch("gen_aniso_kernel");
}

rls_status rls_checksum(rls_context *ctx, int64_t n, const float *data, uint64_t *out_host)
{
    RLS_REQUIRE(ctx != nullptr && out_host != nullptr, "NULL argument");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_REQUIRE(n >= 0, "n < 0");
    *out_host = 0;
    if (n == 0) return RLS_OK;
    RLS_REQUIRE(data != nullptr, "data is NULL");
    RLS_HIP_TRY(hipMemsetAsync(ctx->scratch_u64, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(checksum_kernel, rlsh::grid_for(ctx, n), dim3(rlsh::kBlock), 0, ctx->stream,
                       n, data, ctx->scratch_u64);
    rls_status st = rlsh::check_launch("checksum_kernel");
    if (st != RLS_OK) return st;
    unsigned long long v = 0;
    RLS_HIP_TRY(hipMemcpyAsync(&v, ctx->scratch_u64, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    RLS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out_host = (uint64_t)v;
    return RLS_OK;
}

} // extern "C"
