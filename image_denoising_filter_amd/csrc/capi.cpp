// capi.cpp -- context, error, memory and timer entry points of libmi_denoise.so.
// Replaces the Vulkan bootstrap and buffer factories of the reference
// (src/vk_utils.cpp:13-305, src/main.cpp:247-401) with plain HIP: a context is a device
// plus its four streams (compute, a second compute stream for the frame pipeline, upload, download).
#include "common.hpp"

namespace mid {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

Bind::Bind(mid_ctx *ctx, void *stream) : rc(MID_OK), s(nullptr)
{
    if (!ctx) { rc = set_error(MID_ERR_INVALID, "context is NULL"); return; }
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { rc = set_error(MID_ERR_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e)); return; }
    s = stream ? (hipStream_t)stream : ctx->compute;
}

}  // namespace mid

using namespace mid;

extern "C" const char *mid_last_error(void) { return g_err; }
extern "C" int mid_version(void) { return MID_VERSION; }

extern "C" int mid_ctx_create(int device, mid_ctx **out)
{
    MID_REQUIRE(out != nullptr, "ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(MID_ERR_NO_DEVICE, "no HIP device available (%s): mi_denoise has no CPU path",
                         e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    MID_REQUIRE(device >= 0 && device < n, "ctx_create: device %d outside 0..%d", device, n - 1);
    MID_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MID_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(MID_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
    mid_ctx *c = new mid_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->lds_max = (int)prop.sharedMemPerBlockOptin > 0 ? (int)prop.sharedMemPerBlockOptin : (int)prop.sharedMemPerBlock;
    if (c->lds_max < 160 * 1024 && (int)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024) c->lds_max = 160 * 1024;
    snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    // Streams and hardware queues.  The HIP runtime maps a process's streams onto at most 4 hardware queues per priority level
    // (GPU_MAX_HW_QUEUES), in creation order, a fifth stream sharing the queue of an earlier one; packets of one hardware queue
    // run in order, so two streams that share a queue and both carry waits hold each other back.  Measured on the frame pipeline
    // (64 x 1080p RGBA32F host to host, profiles/r06_pipeline_stream_placement.txt): the four streams created together here,
    // before the process has others: 2858-2868 Mpixel/s; the three pipeline streams created later, by the first pipeline call,
    // so that upload and download landed on ONE queue: 1681; the same late creation with 8 queues per level: 2395-2412; with the
    // copy streams in the high-priority pool: 2527-2564.  Hence: all four are created here, back to back, and the two copy
    // streams -- DMA commands and waits, no workgroups of ours -- at the device's highest priority, which keeps them out of the
    // pool the caller's streams and the kernel streams live in (2849-2856, and 4107-4125 against 4056-4069 for RGBA8 frames).
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    c->copy_priority = greatest;
    if (hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->compute2, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&c->upload, hipStreamNonBlocking, greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&c->download, hipStreamNonBlocking, greatest) != hipSuccess) {
        for (hipStream_t s : {c->compute, c->compute2, c->upload, c->download}) if (s) (void)hipStreamDestroy(s);
        delete c;
        return set_error(MID_ERR_HIP, "ctx_create: stream creation failed");
    }
    *out = c;
    return MID_OK;
}

extern "C" void mid_ctx_destroy(mid_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (hipStream_t s : {ctx->compute, ctx->compute2, ctx->upload, ctx->download}) (void)hipStreamSynchronize(s);
    pipe_cache_release(ctx);
    bounce_release(ctx);
    for (hipStream_t s : {ctx->compute, ctx->compute2, ctx->upload, ctx->download}) (void)hipStreamDestroy(s);
    delete ctx;
}

extern "C" int mid_ctx_release_cached(mid_ctx *ctx)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    std::lock_guard<std::mutex> lock(ctx->pipe.mu);       // (a pipeline call in flight on another thread finishes first)
    pipe_cache_release(ctx);
    bounce_release(ctx);                                   // the two page-locked bounce sets of csrc/hostcopy.cpp (32 MiB)
    return MID_OK;
}

extern "C" int mid_ctx_stream_priorities(mid_ctx *ctx, int priority[4], int *least, int *greatest)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(priority && least && greatest, "ctx_stream_priorities: NULL argument");
    MID_HIP(hipDeviceGetStreamPriorityRange(least, greatest));
    hipStream_t s[4] = {ctx->compute, ctx->compute2, ctx->upload, ctx->download};
    for (int i = 0; i < 4; ++i) MID_HIP(hipStreamGetPriority(s[i], &priority[i]));
    return MID_OK;
}

extern "C" int mid_device_name(mid_ctx *ctx, char *buf, size_t buflen)
{
    MID_REQUIRE(ctx && buf && buflen > 0, "device_name: bad argument");
    snprintf(buf, buflen, "%s", ctx->name);
    return MID_OK;
}

extern "C" int mid_alloc(mid_ctx *ctx, size_t bytes, void **dptr)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(dptr && bytes > 0, "alloc: bad argument");
    MID_HIP(hipMalloc(dptr, bytes));
    return MID_OK;
}

extern "C" int mid_free(mid_ctx *ctx, void *dptr)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    if (dptr) MID_HIP(hipFree(dptr));
    return MID_OK;
}

extern "C" int mid_alloc_host(mid_ctx *ctx, size_t bytes, void **hptr)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(hptr && bytes > 0, "alloc_host: bad argument");
    MID_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return MID_OK;
}

extern "C" int mid_free_host(mid_ctx *ctx, void *hptr)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    if (hptr) MID_HIP(hipHostFree(hptr));
    return MID_OK;
}

extern "C" int mid_host_register(mid_ctx *ctx, void *hptr, size_t bytes)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(hptr && bytes > 0, "host_register: bad argument");
    MID_HIP(hipHostRegister(hptr, bytes, hipHostRegisterPortable));   // pinned for every device of the process
    return MID_OK;
}

extern "C" int mid_host_unregister(mid_ctx *ctx, void *hptr)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(hptr, "host_unregister: NULL pointer");
    MID_HIP(hipHostUnregister(hptr));
    return MID_OK;
}

extern "C" int mid_memcpy_h2d(mid_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(dst && src, "memcpy_h2d: NULL pointer");
    return copy_h2d(ctx, dst, src, bytes, b.s);      // pinned: one async DMA; pageable: through the context's bounce buffers
}

extern "C" int mid_memcpy_d2h(mid_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(dst && src, "memcpy_d2h: NULL pointer");
    return copy_d2h(ctx, dst, src, bytes, b.s);
}

extern "C" int mid_memset(mid_ctx *ctx, void *dst, int value, size_t bytes, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(dst, "memset: NULL pointer");
    if (stream_is_recording(b.s)) return fill_bytes(ctx, dst, value, bytes, b.s);     // (a kernel node instead of a memset node: pointwise.hip says why)
    MID_HIP(hipMemsetAsync(dst, value, bytes, b.s));
    return MID_OK;
}

extern "C" int mid_stream_sync(mid_ctx *ctx, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(b.s, "mid_stream_sync (a host-side wait)")) return rc;
    MID_HIP(hipStreamSynchronize(b.s));
    return MID_OK;
}

// ---- timers -------------------------------------------------------------------------------
struct mid_timer {
    mid_ctx *ctx;
    hipEvent_t a, b;
};

extern "C" int mid_timer_create(mid_ctx *ctx, mid_timer **out)
{
    Bind bd(ctx, nullptr);
    if (bd.rc) return bd.rc;
    MID_REQUIRE(out, "timer_create: out is NULL");
    mid_timer *t = new mid_timer{ctx, nullptr, nullptr};
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) {
        delete t;
        return set_error(MID_ERR_HIP, "timer_create: hipEventCreate failed");
    }
    *out = t;
    return MID_OK;
}

extern "C" int mid_timer_destroy(mid_timer *t)
{
    if (!t) return MID_OK;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return MID_OK;
}

extern "C" int mid_timer_tick(mid_timer *t, void *stream)
{
    MID_REQUIRE(t, "timer_tick: NULL timer");
    Bind b(t->ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(b.s, "mid_timer_tick (its event is read back by the host)")) return rc;
    MID_HIP(hipEventRecord(t->a, b.s));
    return MID_OK;
}

extern "C" int mid_timer_tock(mid_timer *t, void *stream)
{
    MID_REQUIRE(t, "timer_tock: NULL timer");
    Bind b(t->ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(b.s, "mid_timer_tock (its event is read back by the host)")) return rc;
    MID_HIP(hipEventRecord(t->b, b.s));
    return MID_OK;
}

extern "C" int mid_timer_ms(mid_timer *t, float *ms)
{
    MID_REQUIRE(t && ms, "timer_ms: NULL argument");
    MID_HIP(hipEventSynchronize(t->b));
    MID_HIP(hipEventElapsedTime(ms, t->a, t->b));
    return MID_OK;
}
