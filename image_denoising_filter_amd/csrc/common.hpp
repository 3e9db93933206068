// common.hpp -- context, error plumbing and device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_set>
#include "../../include/mi_denoise.h"

struct mid_ctx {
    int device;
    hipStream_t compute;   // default stream for kernels
    hipStream_t compute2;  // second kernel stream of the frame pipeline (consecutive frames alternate)
    hipStream_t upload;    // H2D stream of the frame pipeline
    hipStream_t download;  // D2H stream of the frame pipeline
    int lds_max;           // max dynamic LDS per workgroup (bytes)
    int cu_count;
    char name[128];
    // kernels whose dynamic-LDS limit has been raised on THIS context's device (the attribute is per
    // device, so it is tracked per context; contexts may be used from different threads)
    std::mutex mu;
    std::unordered_set<const void *> lds_configured;
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

// UNORM texel decode, src/texture.cpp:16: c/255 (IEEE-correct division, no fast-math).
__device__ __forceinline__ float4 decode_rgba8(uint32_t v)
{
    return make_float4((float)(v & 0xffu) / 255.0f, (float)((v >> 8) & 0xffu) / 255.0f,
                       (float)((v >> 16) & 0xffu) / 255.0f, (float)(v >> 24) / 255.0f);
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
                                          int x0, int y0, int tid, int nthreads, float rgb_scale = 1.0f)
{
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
        }
    }
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
