// pointwise.hip -- the streaming passes: normalize (shaders/normalize.comp:29-44) and the
// u8 <-> float conversions (src/main.cpp:97-103, :1804-1807; UNORM decode src/texture.cpp:16).
// All three are HBM-bound: one pixel (16 B of float / 4 B of u8) per lane, grid-stride.
#include "common.hpp"

namespace mid {

__global__ __launch_bounds__(256) void normalize_kernel(const mid_weightinfo *__restrict__ W,
                                                        float4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 *wp = (const float4 *)(W + i);
        const float4 wc = wp[0];
        const float nw = wp[1].x;
        float4 o;
        if (nw == 0.0f) o = make_float4(1.0f, 0.0f, 1.0f, 1.0f);                 // normalize.comp:36-38
        else o = make_float4(wc.x / nw, wc.y / nw, wc.z / nw, wc.w / nw);        // :42 (IEEE division)
        out[i] = o;
    }
}

template <int FLAVOUR>
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t *__restrict__ in, float4 *__restrict__ out, size_t npix)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const uint32_t v = in[i];
        const float r = (float)(v & 0xffu), g = (float)((v >> 8) & 0xffu), b = (float)((v >> 16) & 0xffu), a = (float)(v >> 24);
        if (FLAVOUR == 0) out[i] = make_float4(unorm8(r), unorm8(g), unorm8(b), unorm8(a));
        else {
            const float k = 1.0f / 255.0f;                                       // src/main.cpp:1804
            out[i] = make_float4(r * k, g * k, b * k, a * k);
        }
    }
}

template <int FLAVOUR>
__global__ __launch_bounds__(256) void unpack_tail_kernel(const uint8_t *in, float *out, size_t n0, size_t n)
{
    const size_t i = n0 + threadIdx.x;
    if (i < n) out[i] = FLAVOUR == 0 ? unorm8((float)in[i]) : (float)in[i] * (1.0f / 255.0f);
}

__global__ __launch_bounds__(256) void pack_kernel(const float4 *__restrict__ in, uint32_t *__restrict__ out, size_t npix)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const float4 p = in[i];
        out[i] = pack_rgba8(p);
    }
}

__global__ __launch_bounds__(256) void pack_tail_kernel(const float *in, uint8_t *out, size_t n0, size_t n)
{
    const size_t i = n0 + threadIdx.x;
    if (i < n) out[i] = (uint8_t)pack1(in[i]);
}

// mid_memset INSIDE A RECORDING (csrc/recording.cpp) is this kernel, not a captured hipMemsetAsync.  With one build of the library the
// reference's literal multi-frame sequence recorded as [hipMemsetAsync, 5 x nlm_accum, normalize] replayed -- 7 of 7 runs, [-7,7)/[-3,3)
// window, on a created stream and on the context's own -- with the WeightInfo buffer FILLED WITH A PERIODIC PATTERN OF ADDRESS-LIKE WORDS
// where the memset node (the runtime's __amd_rocclr_fillBufferAligned) should have written zeros: a fill whose pattern had been overwritten
// (HIP 7.0.51831 as bundled with torch).  It depends on the process's memory layout: the next build with the memset node restored passed,
// and a stand-alone probe of memset (+ kernel) graphs never shows it (tools/probe_graph_memset.hip, either runtime; LABNOTES R6.10).  A
// kernel node carries its arguments by value, so the clear of a recording is a launch of ours: 16 B per lane, grid-stride, then the
// unaligned head / tail bytes one per lane.
__global__ __launch_bounds__(256) void fill_kernel(uint8_t *dst, uint32_t word, size_t head, size_t n16, size_t tail)
{
    uint4 *body = (uint4 *)(dst + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) body[i] = make_uint4(word, word, word, word);
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) dst[threadIdx.x] = (uint8_t)word;
        if (threadIdx.x < tail) dst[head + n16 * 16 + threadIdx.x] = (uint8_t)word;
    }
}

static unsigned stream_grid(mid_ctx *ctx, size_t n)
{
    // (8 workgroups per CU; 4, 16, 32 and "one pixel per thread" measure the same: profiles/r05_ab_stream_grid_cap.txt)
    const size_t want = (n + 255) / 256, cap = (size_t)ctx->cu_count * 8;
    return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

int fill_bytes(mid_ctx *ctx, void *dst, int value, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MID_OK;
    const uint32_t word = 0x01010101u * (uint32_t)(value & 0xff);
    size_t head = (16 - ((uintptr_t)dst & 15u)) & 15u;
    if (head > bytes) head = bytes;
    const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(ctx, n16)), dim3(256), 0, s, (uint8_t *)dst, word, head, n16, tail);
    MID_HIP(hipGetLastError());
    return MID_OK;
}

}  // namespace mid

using namespace mid;

extern "C" int mid_normalize(mid_ctx *ctx, const mid_normalize_params *p, const mid_weightinfo *W,
                             mid_pixel *out, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(p && p->width > 0 && p->height > 0, "normalize: bad params");
    MID_REQUIRE(W && out, "normalize: NULL pointer");
    MID_REQUIRE((const void *)out != (const void *)W, "normalize: out is the WeightInfo buffer (pixels of different strides would overwrite each other)");
    const size_t n = (size_t)p->width * p->height;
    hipLaunchKernelGGL(normalize_kernel, dim3(stream_grid(ctx, n)), dim3(256), 0, b.s, W, (float4 *)out, n);
    MID_HIP(hipGetLastError());
    return MID_OK;
}

extern "C" int mid_unpack_u8(mid_ctx *ctx, const uint8_t *in, size_t n_values, int flavour, float *out, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(in && out, "unpack_u8: NULL pointer");
    MID_REQUIRE(flavour == 0 || flavour == 1, "unpack_u8: flavour %d is not 0 (UNORM) or 1 (CPU)", flavour);
    MID_REQUIRE(n_values == 0 || (const void *)in != (const void *)out, "unpack_u8: in == out (the conversion is not in place)");
    MID_REQUIRE(((uintptr_t)in & 3u) == 0 && ((uintptr_t)out & 15u) == 0, "unpack_u8: in must be 4-byte and out 16-byte aligned");
    const size_t npix = n_values / 4;
    if (npix) {
        if (flavour == 0) hipLaunchKernelGGL(unpack_kernel<0>, dim3(stream_grid(ctx, npix)), dim3(256), 0, b.s, (const uint32_t *)in, (float4 *)out, npix);
        else              hipLaunchKernelGGL(unpack_kernel<1>, dim3(stream_grid(ctx, npix)), dim3(256), 0, b.s, (const uint32_t *)in, (float4 *)out, npix);
    }
    if (n_values & 3) {
        if (flavour == 0) hipLaunchKernelGGL(unpack_tail_kernel<0>, dim3(1), dim3(256), 0, b.s, in, out, npix * 4, n_values);
        else              hipLaunchKernelGGL(unpack_tail_kernel<1>, dim3(1), dim3(256), 0, b.s, in, out, npix * 4, n_values);
    }
    MID_HIP(hipGetLastError());
    return MID_OK;
}

extern "C" int mid_pack_u8(mid_ctx *ctx, const float *in, size_t n_values, uint8_t *out, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(in && out, "pack_u8: NULL pointer");
    MID_REQUIRE(n_values == 0 || (const void *)in != (const void *)out, "pack_u8: in == out (the conversion is not in place)");
    MID_REQUIRE(((uintptr_t)out & 3u) == 0 && ((uintptr_t)in & 15u) == 0, "pack_u8: out must be 4-byte and in 16-byte aligned");
    const size_t npix = n_values / 4;
    if (npix) hipLaunchKernelGGL(pack_kernel, dim3(stream_grid(ctx, npix)), dim3(256), 0, b.s, (const float4 *)in, (uint32_t *)out, npix);
    if (n_values & 3) hipLaunchKernelGGL(pack_tail_kernel, dim3(1), dim3(256), 0, b.s, in, out, npix * 4, n_values);
    MID_HIP(hipGetLastError());
    return MID_OK;
}
