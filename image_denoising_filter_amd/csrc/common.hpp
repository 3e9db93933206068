// common.hpp -- context, error plumbing and device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_set>
#include <vector>
#include "../../include/mi_denoise.h"

// Device buffers and events of the frame pipeline (csrc/pipeline.cpp), kept from call to call so that a steady stream of
// sequences pays for hipMalloc / hipEventCreate once: grown on demand, released by mid_ctx_release_cached and mid_ctx_destroy.
struct mid_pipe_set { std::vector<void *> p; size_t bytes = 0; };
struct mid_pipe_last { int n_up = 0, nb = 0, f_lo = 0, first = 0, batch = 1; bool direct = false; };   // event layout of the last mid_sequence_nlm* call
struct mid_pipe_cache {
    std::mutex mu;                      // one pipeline call per context at a time: the calls share the context's four streams
    mid_pipe_set ring, out;             // mid_sequence_nlm*: uploaded frames (2k + 4), output slots (4)
    mid_pipe_set target, slots, weights, result;   // mid_nlm_multiframe
    std::vector<hipEvent_t> ev;
    mid_pipe_last last;                 // mid_pipe_last_timeline reads the events of the last call back
};

// Page-locked bounce buffers for host memory the caller did NOT pin (csrc/hostcopy.cpp): two halves used alternately, each
// guarded by the event of the last DMA that read or wrote it.  One set per direction, so the frame pipeline's upload and
// download streams never wait on each other's chunks.  Allocated on the first pageable copy, freed with the context.
struct mid_bounce {
    std::mutex mu;                      // one pageable copy per direction at a time
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};      // ev[i] has been recorded and not yet waited for
    size_t chunk = 0;
};

struct mid_ctx {
    int device;
    // The four streams are created together, in this order, by mid_ctx_create (capi.cpp explains why and at which priorities).
    hipStream_t compute = nullptr;   // default stream for kernels
    hipStream_t compute2 = nullptr;  // second kernel stream of the frame pipeline (consecutive frames alternate)
    hipStream_t upload = nullptr;    // H2D stream of the frame pipeline    -- device's highest stream priority
    hipStream_t download = nullptr;  // D2H stream of the frame pipeline    -- device's highest stream priority
    int copy_priority = 0;           // what upload / download were created with
    int lds_max;           // max dynamic LDS per workgroup (bytes)
    int cu_count;
    char name[128];
    // kernels whose dynamic-LDS limit has been raised on THIS context's device (the attribute is per
    // device, so it is tracked per context; contexts may be used from different threads)
    std::mutex mu;
    std::unordered_set<const void *> lds_configured;
    mid_pipe_cache pipe;
    mid_bounce bounce_up, bounce_down;   // pageable host memory never reaches hipMemcpyAsync: see csrc/hostcopy.cpp
};

namespace mid {

int set_error(int code, const char *fmt, ...);

#define MID_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return mid::set_error(MID_ERR_HIP, "%s failed: %s (%s:%d)", #call,          \
                                  hipGetErrorString(e__), __FILE__, __LINE__);          \
    } while (0)

#define MID_REQUIRE(cond, ...)                                                          \
    do {                                                                                \
        if (!(cond)) return mid::set_error(MID_ERR_INVALID, __VA_ARGS__);               \
    } while (0)

// Binds the calling thread to the context's device and resolves the stream argument.
struct Bind {
    int rc;
    hipStream_t s;
    Bind(mid_ctx *ctx, void *stream);
};

inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

// Entry points that wait on the host, allocate, or drive several streams cannot be part of a recording (csrc/recording.cpp: a stream
// between mid_record_begin and mid_record_end is in HIP's capture mode): they refuse with a message instead of leaving an invalidated
// capture behind.
inline int refuse_if_recording(hipStream_t s, const char *what)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return MID_OK; }
    if (st != hipStreamCaptureStatusNone)
        return set_error(MID_ERR_INVALID, "%s cannot be part of a recording (mid_record_begin is open on this stream): only the calls that "
                         "enqueue on ONE stream without waiting can -- kernels, mid_memset, copies from / to page-locked memory", what);
    return MID_OK;
}

// Frees what the frame pipeline keeps in the context (pipeline.cpp); the context's streams must be idle.
void pipe_cache_release(mid_ctx *ctx);

// Host <-> device copies that never hand the runtime pageable memory (csrc/hostcopy.cpp).  Pinned host memory
// (hipHostMalloc / hipHostRegister: mid_alloc_host, mid_host_register, mid_image_load_pinned) is DMA'd in place and the
// call is asynchronous on `s`; anything else goes through the context's page-locked bounce buffers in chunks: copy_h2d
// returns when the source has been consumed (the last DMAs may still be in flight on `s`), copy_d2h when the data is in dst.
bool host_is_pinned(const void *p, size_t bytes);
int copy_h2d(mid_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t s);
int copy_d2h(mid_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t s);
void bounce_release(mid_ctx *ctx);     // waits for the last chunks and frees both bounce sets

// memset as a kernel launch (pointwise.hip): what mid_memset enqueues while its stream records -- see there why.
int fill_bytes(mid_ctx *ctx, void *dst, int value, size_t bytes, hipStream_t s);
inline bool stream_is_recording(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// mid_nlm_temporal with the output format as an argument: out_u8 != 0 writes RGBA8 frames (pack_rgba8 of the
// normalized pixel) instead of float4 ones -- used by the frame pipeline's u8 variant, not exported.
// `corunning` != 0: the caller keeps launches on two streams in flight (the frame pipeline), so the last round of one launch
// overlaps the first of the next -- the HALF launch shape for a small launch's last round (nlm.hip, tail_split) is not used.
int nlm_temporal_out(mid_ctx *ctx, const mid_nlm_params *p, const void *const *frames, int n_frames, int k,
                     int first, int count, void *const *out, int out_u8, void *stream, int corunning = 0);

// ROCTx ranges (csrc/markers.cpp): no-ops unless the process already holds a ROCTx (rocprofv3 --marker-trace preloads one).
bool markers_active();
void range_push(const char *name);
void range_pop();
struct Range {
    bool on;
    explicit Range(const char *fmt, ...) __attribute__((format(printf, 2, 3)));
    ~Range();
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

// Raise the dynamic-LDS limit of `kern` once per context (kernels here use up to 160 KB).
inline int ensure_lds(mid_ctx *ctx, const void *kern, size_t bytes)
{
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->lds_configured.count(kern)) return MID_OK;
    MID_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->lds_configured.insert(kern);
    return MID_OK;
}

// ---- device side ------------------------------------------------------------------------
// Frame table passed by value to the batched kernels (kernarg space, scalar loads).
constexpr int kMaxFrames = 96;
struct FrameTable { const void *p[kMaxFrames]; };
struct OutTable   { void *p[kMaxFrames]; };

// UNORM texel decode, src/texture.cpp:16: the correctly rounded quotient c/255 for c = 0..255.  A product with
// 1/255 alone is off by one ulp for 126 of the 256 codes; one residual correction (r = c - 255 q exactly, q += r/255)
// lands on the IEEE quotient for all 256 -- checked exhaustively in exact arithmetic and by the bit-exact u8 tests --
// in 3 instructions instead of the ~12 plus two mode switches of a full fp32 division.
__device__ __forceinline__ float unorm8(float c)
{
    const float k = 1.0f / 255.0f;
    const float q = c * k;
    return fmaf(fmaf(-q, 255.0f, c), k, q);
}
__device__ __forceinline__ float4 decode_rgba8(uint32_t v)
{
    return make_float4(unorm8((float)(v & 0xffu)), unorm8((float)((v >> 8) & 0xffu)),
                       unorm8((float)((v >> 16) & 0xffu)), unorm8((float)(v >> 24)));
}

// u8 encode of the reference's read-back, src/main.cpp:97-103: (unsigned char)(255.0f * v), truncation, all four
// channels; clamped only where that C cast is undefined (v <= -1, v >= 256, NaN).
__device__ __forceinline__ uint32_t pack1(float x)
{
    const float v = 255.0f * x;                       // src/main.cpp:99
    if (!(v > -1.0f)) return 0u;                      // C cast undefined (and NaN): clamp
    if (v >= 256.0f) return 255u;
    return (uint32_t)(int)v;                          // truncation toward zero
}
__device__ __forceinline__ uint32_t pack_rgba8(float4 p)
{
    return pack1(p.x) | (pack1(p.y) << 8) | (pack1(p.z) << 16) | (pack1(p.w) << 24);
}

// 2-D fetch with the zero-texel policy for out-of-image coordinates (texelFetch of the
// sampler2D shaders; SURVEY.md 8a: OOB = vec4(0)).
template <int FMT>
__device__ __forceinline__ float4 fetch_texture(const void *img, int w, int h, int x, int y)
{
    if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t idx = (size_t)y * w + x;
    if (FMT == MID_FMT_RGBA8) return decode_rgba8(((const uint32_t *)img)[idx]);
    return ((const float4 *)img)[idx];
}

// Flat fetch of the samplerBuffer shader (bialteral_linear.comp:58): index y*w + x with x
// possibly outside the row, so it lands in the adjacent row; outside [0,N) = vec4(0).
template <int FMT>
__device__ __forceinline__ float4 fetch_linear(const void *img, int w, int h, int x, int y)
{
    const long idx = (long)y * w + x;
    if (idx < 0 || idx >= (long)w * h) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (FMT == MID_FMT_RGBA8) return decode_rgba8(((const uint32_t *)img)[idx]);
    return ((const float4 *)img)[idx];
}

// Cooperative fill of an LDS tile of tw x th texels whose top-left texel is image (x0,y0).
// Consecutive threads take consecutive texels of a tile row: 16 B/lane coalesced HBM reads.
template <int FMT, bool LINEAR>
__device__ __forceinline__ void fill_tile(float4 *lds, int tw, int th, const void *img, int w, int h,
                                          int x0, int y0, int tid, int nthreads, float rgb_scale = 1.0f, bool *opaque = nullptr)
{
    // `opaque` (optional): and-ed with "every texel THIS thread stored has alpha == 1.0f" (out-of-image texels are vec4(0): not opaque)
    // four texels per thread per trip: the four global loads are in flight together, so a tile costs
    // about n/(4*nthreads) memory latencies instead of n/nthreads
    const int n = tw * th;
    for (int t0 = tid; t0 < n; t0 += 4 * nthreads) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * nthreads;
            const int ty = t / tw, tx = t - ty * tw;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < n) v[j] = LINEAR ? fetch_linear<FMT>(img, w, h, x0 + tx, y0 + ty)
                                     : fetch_texture<FMT>(img, w, h, x0 + tx, y0 + ty);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * nthreads;
            // rgb_scale is 1 except for the NLM strip kernels (exponent scale folded into the colours);
            // x * 1.0f is exact, and alpha is never scaled
            if (t < n) lds[t] = make_float4(v[j].x * rgb_scale, v[j].y * rgb_scale, v[j].z * rgb_scale, v[j].w);
            if (opaque && t < n) *opaque = *opaque && v[j].w == 1.0f;
        }
    }
}

// XCD-aware remap of a workgroup's tile index INSIDE its frame.  Workgroups are dealt round-robin over
// the 8 XCDs (XCD = linear id % 8).  Frames stay in launch order -- every XCD gets an equal share of every
// output frame, which matters because frames at the ends of a sequence have shorter temporal windows
// (an earlier whole-grid remap gave XCD 0 only 3-neighbour frames and XCD 3 only 5-neighbour ones: 18 %
// slower) -- and within a frame the tiles an XCD receives are made one contiguous run, so neighbouring
// tiles share halo texels in that XCD's L2.  Bijective for any tile count; speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap_in_frame(unsigned t, unsigned tiles, unsigned frame)
{
    const unsigned off = (frame * tiles) & 7u;          // XCD of this frame's tile 0
    const unsigned c = (t + off) & 7u;                  // XCD this workgroup runs on
    unsigned start = 0;                                 // tiles owned by XCDs before c
    for (unsigned cc = 0; cc < c; ++cc) {
        const unsigned first = (cc + 8u - off) & 7u;
        start += first < tiles ? (tiles - first + 7u) >> 3 : 0u;
    }
    const unsigned first_c = (c + 8u - off) & 7u;
    return start + ((t - first_c) >> 3);
}

// 2^x on the transcendental pipe.  On gfx950 a v_exp_f32 that is followed IMMEDIATELY by another vector instruction of the
// same wave costs about 7 cycles more than its own 8 (tools/microbench10/11.hip: 8 exps + 88 FMAs take 357 cycles per group per
// SIMD issued back to back, 302 with one wait state -- or any scalar instruction -- after each exp; the parts alone sum to 278).
// Issuing the instruction from here with its wait state attached keeps the pair together through scheduling.  Same
// instruction, same result bits as __builtin_amdgcn_exp2f.  Used by the bilateral kernels' single taps (measured +3.5 % on the
// tap-by-tap loop, profiles/r03_ab_exp_wait_state.txt); loops that issue their exps as a burst at raised priority -- the tiled
// bilateral kernel's row groups, the NLM offset loop -- use the builtin (there the wait states measured -2 % / -6 %).
__device__ __forceinline__ float exp2_hw(float x)
{
    float r;
    asm("v_exp_f32_e32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(x));
    return r;
}
// Whole-wave lane shifts through DPP (no LDS traffic): value of lane l-1 / l+1; lanes without
// a source read 0.
__device__ __forceinline__ float wave_shr1(float v)   // result[l] = v[l-1]
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v)   // result[l] = v[l+1]
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

}  // namespace mid
