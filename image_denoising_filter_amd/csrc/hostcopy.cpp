// hostcopy.cpp -- every host <-> device copy of the library goes through here.
//
// The reference only ever copies from / to memory it mapped for the device itself: LoadImageDataToBuffer memcpy's the
// decoded pixels into a host-visible staging buffer (src/main.cpp:1105-1142) and GetImageFromGPU reads one back
// (:91-123).  The HIP counterpart of that staging memory is page-locked host memory (hipHostMalloc / hipHostRegister).
// A C-ABI cannot stop a caller from passing an ordinary malloc'd frame, though, and what the HIP runtime does with one
// is NOT a plain copy: for more than 1 MiB (ROCclr pinnedMinXferSize_) it page-locks the caller's pages on the fly in
// 32 MiB windows (hsa_amd_memory_lock_to_pool: a KFD userptr mapping per window), lets the DMA engine or a blit kernel
// read them in place, waits, and unlocks -- per copy.  Round 4's one unexplained abort of the GPU suite was raised on a
// runtime thread while the main thread sat in exactly such a copy (LABNOTES R5.1), so the product no longer depends on
// that path: pageable memory is moved through page-locked bounce buffers owned by the context, in chunks, by this file,
// and hipMemcpyAsync only ever sees pinned host pointers.
//
// Cost, so that callers can choose: a pinned source is one asynchronous DMA at the link's rate; a pageable one adds a
// host memcpy per chunk (overlapped with the previous chunk's DMA) and the call does not return before the caller's
// buffer has been read (h2d) or filled (d2h) -- the same completion rule hipMemcpyAsync itself applies to pageable memory.
#include "common.hpp"

namespace mid {

namespace {

constexpr size_t kChunk = 8u << 20;      // two 8 MiB halves per direction: a chunk's DMA (~0.17 ms) hides under the next memcpy

void bounce_drop(mid_bounce &b)           // caller holds b.mu
{
    for (int i = 0; i < 2; ++i) {
        if (b.busy[i]) { (void)hipEventSynchronize(b.ev[i]); b.busy[i] = false; }
        if (b.ev[i]) { (void)hipEventDestroy(b.ev[i]); b.ev[i] = nullptr; }
        if (b.buf[i]) { (void)hipHostFree(b.buf[i]); b.buf[i] = nullptr; }
    }
    b.chunk = 0;
}

int bounce_prepare(mid_bounce &b)
{
    if (b.chunk) return MID_OK;             // (set last: a half-built set is never mistaken for a complete one)
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipHostMalloc(&b.buf[i], kChunk, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b.ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            bounce_drop(b);
            return set_error(MID_ERR_HIP, "page-locked bounce buffer (%zu B): %s", kChunk, hipGetErrorString(e));
        }
        b.busy[i] = false;
    }
    b.chunk = kChunk;
    return MID_OK;
}

int bounce_wait(mid_bounce &b, int i)
{
    if (b.busy[i]) {
        MID_HIP(hipEventSynchronize(b.ev[i]));
        b.busy[i] = false;
    }
    return MID_OK;
}

void bounce_free(mid_bounce &b)
{
    std::lock_guard<std::mutex> lock(b.mu);
    bounce_drop(b);
}

// hipMemoryTypeHost covers hipHostMalloc and hipHostRegister memory alike; a pointer the runtime has never seen comes
// back as hipMemoryTypeUnregistered (ROCm >= 6) or as hipErrorInvalidValue (older), which must not stay behind as the
// thread's "last error".
enum class Kind { Pageable, Pinned, Device };

Kind kind_of(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return Kind::Pageable; }
    if (a.type == hipMemoryTypeHost) return Kind::Pinned;
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) return Kind::Device;
    return Kind::Pageable;                        // unregistered, or managed memory (host-addressable: a plain memcpy works)
}

// Both ends of the range must lie in pinned memory (a frame that straddles the end of a registered range is treated as
// pageable: bounced, never handed to the runtime half-pinned).  A device pointer in the place of the host buffer would be
// memcpy'd by the CPU in the bounce path: refused instead of crashing.
int classify(const void *host, size_t bytes, const char *what, bool *pinned)
{
    const Kind k0 = kind_of(host);
    if (k0 == Kind::Device) return set_error(MID_ERR_INVALID, "%s: the host-side pointer %p is device memory", what, host);
    *pinned = k0 == Kind::Pinned && kind_of((const char *)host + bytes - 1) == Kind::Pinned;
    return MID_OK;
}

}  // namespace

bool host_is_pinned(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return true;
    return kind_of(p) == Kind::Pinned && kind_of((const char *)p + bytes - 1) == Kind::Pinned;
}

int copy_h2d(mid_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MID_OK;
    bool pinned = false;
    if (int rc = classify(src, bytes, "host-to-device copy", &pinned)) return rc;
    if (pinned) {
        MID_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return MID_OK;
    }
    if (int rc = refuse_if_recording(s, "a copy from pageable host memory (bounced with host-side waits; page-lock it: mid_alloc_host / mid_host_register)")) return rc;
    mid_bounce &b = ctx->bounce_up;
    std::lock_guard<std::mutex> lock(b.mu);
    if (int rc = bounce_prepare(b)) return rc;
    int i = 0;
    for (size_t off = 0; off < bytes; off += b.chunk, i ^= 1) {
        const size_t n = bytes - off < b.chunk ? bytes - off : b.chunk;
        if (int rc = bounce_wait(b, i)) return rc;                   // the DMA that last read this half has finished
        memcpy(b.buf[i], (const char *)src + off, n);
        MID_HIP(hipMemcpyAsync((char *)dst + off, b.buf[i], n, hipMemcpyHostToDevice, s));
        MID_HIP(hipEventRecord(b.ev[i], s));
        b.busy[i] = true;
    }
    return MID_OK;                                                   // src is consumed; the halves stay guarded by their events
}

int copy_d2h(mid_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MID_OK;
    bool pinned = false;
    if (int rc = classify(dst, bytes, "device-to-host copy", &pinned)) return rc;
    if (pinned) {
        MID_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        return MID_OK;
    }
    if (int rc = refuse_if_recording(s, "a copy to pageable host memory (bounced with host-side waits; page-lock it: mid_alloc_host / mid_host_register)")) return rc;
    mid_bounce &b = ctx->bounce_down;
    std::lock_guard<std::mutex> lock(b.mu);
    if (int rc = bounce_prepare(b)) return rc;
    // chunk c is in flight into half c & 1 while chunk c-1 is copied out of the other half
    size_t prev_off = 0, prev_n = 0;
    int i = 0;
    for (size_t off = 0; off < bytes; off += b.chunk, i ^= 1) {
        const size_t n = bytes - off < b.chunk ? bytes - off : b.chunk;
        if (int rc = bounce_wait(b, i)) return rc;                   // (only ever pending after an error path of an earlier call)
        MID_HIP(hipMemcpyAsync(b.buf[i], (const char *)src + off, n, hipMemcpyDeviceToHost, s));
        MID_HIP(hipEventRecord(b.ev[i], s));
        b.busy[i] = true;
        if (prev_n) {
            if (int rc = bounce_wait(b, i ^ 1)) return rc;
            memcpy((char *)dst + prev_off, b.buf[i ^ 1], prev_n);
        }
        prev_off = off; prev_n = n;
    }
    if (int rc = bounce_wait(b, i ^ 1)) return rc;
    memcpy((char *)dst + prev_off, b.buf[i ^ 1], prev_n);
    return MID_OK;
}

void bounce_release(mid_ctx *ctx)
{
    bounce_free(ctx->bounce_up);
    bounce_free(ctx->bounce_down);
}

}  // namespace mid
