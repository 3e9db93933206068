// microbench2.hip -- ds_bpermute_b32 throughput alone and beside VALU work (can the LDS crossbar
// take over part of the cross-lane box sum from the half-rate DPP adds?).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const int lane = threadIdx.x & 63;
    const int addr = ((lane + 3) & 63) * 4;
    const float b = seed * 0.5f, c = seed * 0.25f;
    __shared__ float4 sm[1024];
    if (KIND == 3) { for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = make_float4(seed, 1.f, 2.f, 3.f); __syncthreads(); }
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) {          // 16 bpermutes
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[i])));
        } else if (KIND == 1) {   // 4 bpermutes + 48 fma (ratio like the proposed NLM: 16 bperm per ~190 VALU)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[i])));
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 2) {   // 48 fma only
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 3) {   // 4 bpermutes + 4 ds_read_b128 + 48 fma
            const float4 *p = sm + lane + ((it & 7) << 6);
#pragma unroll
            for (int i = 0; i < 4; ++i) { float4 v = p[(i & 3) * 128]; asm volatile("" :: "v"(v.y), "v"(v.z), "v"(v.w)); a[8 + i] += v.x; }
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[i])));
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 4) {   // 16 ds_swizzle (rotate by 3 within 32)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[i]), 0xC060));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(const char *name, int wps)
{
    const int blocks = 256 * wps;
    float *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/SIMD=%d  %.3f ms  %.1f clk(2.4GHz)/iter/SIMD-wave\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)wps * ITERS));
    CK(hipFree(d));
    return 0;
}

int main()
{
    for (int w : {2, 4}) {
        run<0>("16 ds_bpermute_b32", w);
        run<4>("16 ds_swizzle_b32", w);
        run<2>("48 v_fma", w);
        run<1>("4 bpermute + 48 v_fma", w);
        run<3>("4 bpermute + 4 ds_read_b128 + 48 v_fma", w);
    }
    return 0;
}
