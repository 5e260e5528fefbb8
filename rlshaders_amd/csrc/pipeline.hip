// pipeline.hip -- host-resident batches: pinned host memory and a chunked, overlapped
// host -> device -> kernel -> device -> host pipeline (rls_pipeline_*), host code only.
//
// Why: the reference's closures run per hit on CPU render threads (src/rlGgx.cpp:248-261 builds the
// closure on the stack of shader_evaluate), so the shading points an Arnold-side stub gathers start in
// HOST memory and the results are wanted there.  A batch that is uploaded, processed and downloaded in
// three serial steps leaves the GPU idle during both copies and each DMA direction idle during the other;
// here the batch is cut into chunks that travel through `depth` slots, each slot a stream of its own with
// its own device planes: the upload of chunk k + 1, the kernels of chunk k and the download of chunk k - 1
// run concurrently (the two copy directions on separate DMA engines).  Within a slot everything is in
// stream order, so a slot's buffers are reused without events.
//
// Bound: PCIe, not HBM and not the kernels -- config 2 moves 76 B up and 48 B down per point.
#include <stdlib.h>
#include <string.h>

#include "rls_internal.hpp"

namespace {

struct Slot {
    rls_context *ctx;          // a context of its own = a stream of its own, same device
    float *block;              // (in_planes + out_planes) planes of `stride` floats
    float **in, **out;         // plane pointers handed to the launch callback
};

// host planes [k0, k1) <-> device planes of one slot, rows [p0, p0 + count): one hipMemcpyAsync per plane, or -- where
// consecutive host planes are a constant positive distance apart -- one hipMemcpy2DAsync per run of planes
// whose spacing is at least the batch's n points, i.e. planes that do not overlap.  (A run may be equally spaced by accident
// -- separate allocations next to each other -- and then the runtime refuses the strided copy, whose source must lie in one
// allocation: the run goes plane by plane, and so does the rest of THIS rls_pipeline_run call; `strided` is the caller's
// per-run flag, the next run looks at its own planes afresh.)
template <class HostPtr>
hipError_t copy_planes(bool up, int planes, HostPtr const *host, float *const *dev, int64_t dev_stride, int64_t p0,
                       int64_t count, int64_t n, hipStream_t stream, bool &strided)
{
    hipError_t e = hipSuccess;
    const size_t row = (size_t)count * sizeof(float);
    for (int k = 0; e == hipSuccess && k < planes;) {
        if (!host[k]) { k++; continue; }
        int m = k + 1;                                                // the run [k, m) of equally spaced host planes
        if (strided && m < planes && host[m] && host[m] > host[k]) {
            const ptrdiff_t d = host[m] - host[k];
            if (d < n) { m = k + 1; goto single; }                        // closer than n floats: not the planes of one [planes, n] array
            while (m + 1 < planes && host[m + 1] && host[m + 1] - host[m] == d) m++;
            m++;
            if (m - k >= 2) {
                if (up) e = hipMemcpy2DAsync(dev[k], (size_t)dev_stride * sizeof(float), host[k] + p0, (size_t)d * sizeof(float), row,
                                             (size_t)(m - k), hipMemcpyHostToDevice, stream);
                else e = hipMemcpy2DAsync((void *)(host[k] + p0), (size_t)d * sizeof(float), dev[k], (size_t)dev_stride * sizeof(float), row,
                                          (size_t)(m - k), hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) { k = m; continue; }
                (void)hipGetLastError();                                  // refused (arguments are validated before anything is queued)
                e = hipSuccess;
                strided = false;
            }
            m = k + 1;
        }
    single:
        if (up) e = hipMemcpyAsync(dev[k], host[k] + p0, row, hipMemcpyHostToDevice, stream);
        else e = hipMemcpyAsync((void *)(host[k] + p0), dev[k], row, hipMemcpyDeviceToHost, stream);
        k = m;
    }
    return e;
}

} // namespace

struct rls_pipeline {
    rls_context *parent;
    int64_t chunk_points, stride;
    int in_planes, out_planes, depth;
    bool strided;              // RLS_PIPELINE_STRIDED=0 turns the strided copies off (one copy per plane); fixed at creation
    Slot *slots;
};

extern "C" {

rls_status rls_host_alloc(rls_context *ctx, size_t bytes, void **out)
{
    RLS_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    *out = nullptr;
    if (bytes == 0) return RLS_OK;
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return RLS_OK;
}

rls_status rls_host_free(rls_context *ctx, void *p)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!p) return RLS_OK;
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipHostFree(p));
    return RLS_OK;
}

rls_status rls_host_register(rls_context *ctx, void *p, size_t bytes)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    RLS_REQUIRE(p != nullptr && bytes > 0, "empty range");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return RLS_OK;
}

rls_status rls_host_unregister(rls_context *ctx, void *p)
{
    RLS_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!p) return RLS_OK;
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    RLS_HIP_TRY(hipHostUnregister(p));
    return RLS_OK;
}

void rls_pipeline_destroy(rls_pipeline *p)
{
    if (!p) return;
    for (int s = 0; p->slots && s < p->depth; s++) {
        Slot &sl = p->slots[s];
        if (sl.ctx) {
            (void)hipSetDevice(sl.ctx->device);
            (void)hipStreamSynchronize(sl.ctx->stream);
        }
        if (sl.block) (void)hipFree(sl.block);
        free(sl.in);
        free(sl.out);
        if (sl.ctx) rls_context_destroy(sl.ctx);
    }
    free(p->slots);
    free(p);
}

rls_status rls_pipeline_create(rls_context *ctx, int64_t chunk_points, int in_planes, int out_planes, int depth,
                               rls_pipeline **out)
{
    RLS_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    *out = nullptr;
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_REQUIRE(chunk_points >= 1, "chunk_points < 1");
    RLS_REQUIRE(in_planes >= 0 && out_planes >= 0 && in_planes + out_planes >= 1 && in_planes + out_planes <= 4096,
                "plane counts out of range");
    RLS_REQUIRE(depth >= 1 && depth <= 16, "depth must be in [1, 16]");
    rls_pipeline *p = (rls_pipeline *)calloc(1, sizeof(rls_pipeline));
    if (!p) { rlsh::set_error("rls_pipeline_create: out of host memory"); return RLS_ERR_OUT_OF_MEMORY; }
    p->parent = ctx;
    p->chunk_points = chunk_points;
    p->stride = (chunk_points + 63) / 64 * 64;              // planes on 256-byte boundaries
    p->in_planes = in_planes; p->out_planes = out_planes; p->depth = depth;
    {
        const char *sv = getenv("RLS_PIPELINE_STRIDED");
        p->strided = !(sv && sv[0] == '0');
    }
    p->slots = (Slot *)calloc((size_t)depth, sizeof(Slot));
    rls_status st = p->slots ? RLS_OK : RLS_ERR_OUT_OF_MEMORY;
    for (int s = 0; st == RLS_OK && s < depth; s++) {
        Slot &sl = p->slots[s];
        st = rls_context_create(ctx->device, &sl.ctx);
        if (st != RLS_OK) break;
        sl.ctx->blocks_per_cu = ctx->blocks_per_cu;
        const size_t planes = (size_t)(in_planes + out_planes);
        hipError_t e = hipMalloc((void **)&sl.block, planes * (size_t)p->stride * sizeof(float));
        if (e != hipSuccess) { st = rlsh::hip_fail(e, "rls_pipeline_create: hipMalloc"); break; }
        sl.in = (float **)calloc((size_t)(in_planes > 0 ? in_planes : 1), sizeof(float *));
        sl.out = (float **)calloc((size_t)(out_planes > 0 ? out_planes : 1), sizeof(float *));
        if (!sl.in || !sl.out) { st = RLS_ERR_OUT_OF_MEMORY; break; }
        for (int k = 0; k < in_planes; k++) sl.in[k] = sl.block + (size_t)k * (size_t)p->stride;
        for (int k = 0; k < out_planes; k++) sl.out[k] = sl.block + (size_t)(in_planes + k) * (size_t)p->stride;
    }
    if (st != RLS_OK) {
        if (st == RLS_ERR_OUT_OF_MEMORY) rlsh::set_error("rls_pipeline_create: out of memory");
        rls_pipeline_destroy(p);
        return st;
    }
    *out = p;
    return RLS_OK;
}

rls_status rls_pipeline_run(rls_pipeline *p, int64_t n, const float *const *host_in, float *const *host_out,
                            rls_pipeline_launch_fn launch, void *user)
{
    RLS_REQUIRE(p != nullptr, "pipeline is NULL");
    RLS_REQUIRE(n >= 0, "n < 0");
    RLS_REQUIRE(launch != nullptr, "launch callback is NULL");
    RLS_REQUIRE(p->in_planes == 0 || host_in != nullptr, "host_in is NULL");
    RLS_REQUIRE(p->out_planes == 0 || host_out != nullptr, "host_out is NULL");
    RLS_REQUIRE(!p->parent->capturing, "not allowed while a launch graph is being recorded");
    rls_status st = RLS_OK;
    int64_t c = 0;
    bool strided = p->strided;             // this run's: a refused strided copy sends the rest of THIS run plane by plane
    for (int64_t p0 = 0; p0 < n && st == RLS_OK; p0 += p->chunk_points, c++) {
        const int64_t count = n - p0 < p->chunk_points ? n - p0 : p->chunk_points;
        Slot &sl = p->slots[c % p->depth];
        sl.ctx->fast = p->parent->fast;                    // the slots compute in the parent's arithmetic mode
        hipStream_t stream = sl.ctx->stream;
        hipError_t e = hipSetDevice(sl.ctx->device);
        if (e != hipSuccess) { st = rlsh::hip_fail(e, "rls_pipeline_run: hipSetDevice"); break; }
        // a NULL host plane is a plane the caller does not stream (a uniform parameter, an unwanted output).  Runs of host
        // planes that are equally spaced in memory (rlsb::HostPlanes, a [planes, n] array) travel as ONE strided copy
        e = copy_planes(true, p->in_planes, host_in, sl.in, p->stride, p0, count, n, stream, strided);
        if (e != hipSuccess) { st = rlsh::hip_fail(e, "rls_pipeline_run: upload"); break; }
        st = launch(user, sl.ctx, p0, count, sl.in, sl.out);
        if (st != RLS_OK) break;
        e = copy_planes(false, p->out_planes, host_out, sl.out, p->stride, p0, count, n, stream, strided);
        if (e != hipSuccess) { st = rlsh::hip_fail(e, "rls_pipeline_run: download"); break; }
    }
    // drain every slot, also after a failure: the caller's host buffers must not be written behind its back
    for (int s = 0; s < p->depth; s++) {
        hipError_t e = hipStreamSynchronize(p->slots[s].ctx->stream);
        if (e != hipSuccess && st == RLS_OK) st = rlsh::hip_fail(e, "rls_pipeline_run: synchronize");
    }
    return st;
}

// Pinned-memory copy rates of this box, GB/s: [0] host -> device alone, [1] device -> host alone, [2] both directions at
// once (sum of the two), each over `bytes` per direction in 64 MiB pieces on two streams -- what a pipeline's copies can
// reach at best, measured where the pipeline runs.
rls_status rls_measure_copy_rates(rls_context *ctx, size_t bytes, float rates_gb_per_s[3])
{
    RLS_REQUIRE(ctx != nullptr && rates_gb_per_s != nullptr, "NULL argument");
    RLS_REQUIRE(!ctx->capturing, "not allowed while a launch graph is being recorded");
    RLS_REQUIRE(bytes >= (1u << 20), "bytes < 1 MiB");
    RLS_HIP_TRY(hipSetDevice(ctx->device));
    void *h0 = nullptr, *h1 = nullptr, *d0 = nullptr, *d1 = nullptr;
    hipStream_t s0 = nullptr, s1 = nullptr;
    hipEvent_t a = nullptr, b = nullptr, c = nullptr;
    hipError_t e = hipHostMalloc(&h0, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&h1, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d0, bytes);
    if (e == hipSuccess) e = hipMalloc(&d1, bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) e = hipEventCreate(&c);
    if (e == hipSuccess) {
        memset(h0, 1, bytes);
        memset(h1, 2, bytes);
        const size_t piece = 64u << 20;       // large pieces: what a copy engine sustains (4 MiB pieces read ~10 % lower)
        auto pass = [&](bool up, bool down, float *gbs) -> hipError_t {
            hipError_t r = hipDeviceSynchronize();
            if (r == hipSuccess) r = hipEventRecord(a, s0);
            if (r == hipSuccess) r = hipStreamWaitEvent(s1, a, 0);
            for (size_t off = 0; r == hipSuccess && off < bytes; off += piece) {
                const size_t m = bytes - off < piece ? bytes - off : piece;
                if (up) r = hipMemcpyAsync((char *)d0 + off, (char *)h0 + off, m, hipMemcpyHostToDevice, s0);
                if (r == hipSuccess && down) r = hipMemcpyAsync((char *)h1 + off, (char *)d1 + off, m, hipMemcpyDeviceToHost, s1);
            }
            if (r == hipSuccess) r = hipEventRecord(b, s1);
            if (r == hipSuccess) r = hipStreamWaitEvent(s0, b, 0);
            if (r == hipSuccess) r = hipEventRecord(c, s0);
            if (r == hipSuccess) r = hipEventSynchronize(c);
            float ms = 0.0f;
            if (r == hipSuccess) r = hipEventElapsedTime(&ms, a, c);
            if (r == hipSuccess) *gbs = (float)((double)bytes * ((up ? 1 : 0) + (down ? 1 : 0)) / (ms * 1e-3) / 1e9);
            return r;
        };
        float warm;
        e = pass(true, true, &warm);
        rates_gb_per_s[0] = rates_gb_per_s[1] = rates_gb_per_s[2] = 0.0f;
        for (int rep = 0; e == hipSuccess && rep < 2; rep++) {          // best of two: a pass now and then runs at half rate
            float g[3] = {0.0f, 0.0f, 0.0f};
            e = pass(true, false, &g[0]);
            if (e == hipSuccess) e = pass(false, true, &g[1]);
            if (e == hipSuccess) e = pass(true, true, &g[2]);
            for (int k = 0; k < 3; k++) if (g[k] > rates_gb_per_s[k]) rates_gb_per_s[k] = g[k];
        }
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (c) (void)hipEventDestroy(c);
    if (s0) (void)hipStreamDestroy(s0);
    if (s1) (void)hipStreamDestroy(s1);
    if (d0) (void)hipFree(d0);
    if (d1) (void)hipFree(d1);
    if (h0) (void)hipHostFree(h0);
    if (h1) (void)hipHostFree(h1);
    if (e != hipSuccess) return rlsh::hip_fail(e, "rls_measure_copy_rates");
    return RLS_OK;
}

} // extern "C"
