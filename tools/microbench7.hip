// microbench7.hip -- does an f16 MFMA (v_mfma_f32_16x16x16_f16) overlap with VALU work on gfx950?  (microbench3 showed that
// an f32 MFMA does not: 160 FMAs + 8 MFMAs took the sum of their separate times.  If the f16 matrix pipe runs beside
// the vector pipe, the NLM kernel's horizontal box sums could move there as 0/1 band products of f16 hi/lo splits.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int NFMA, int NMFMA>
__global__ __launch_bounds__(256) void probe(float *out, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const float b = seed * 0.5f, c = seed * 0.25f;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const h4 wa = {(_Float16)((threadIdx.x & 3) ? 1.f : 0.f), (_Float16)1.f, (_Float16)0.f, (_Float16)1.f};
    h4 vb = {(_Float16)seed, (_Float16)(seed * 2), (_Float16)(seed * 3), (_Float16)(threadIdx.x & 7)};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa, vb, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NFMA / 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NFMA, int NMFMA>
int run(const char *name, int wps)
{
    const int blocks = 256 * wps;
    float *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<NFMA, NMFMA>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<NFMA, NMFMA>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/SIMD=%d  %.3f ms  %.1f clk(2.4GHz)/iter/SIMD-wave\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)wps * ITERS));
    CK(hipFree(d));
    return 0;
}

int main()
{
    for (int w : {2, 4}) {
        run<160, 0>("160 fma", w);
        run<0, 8>("8 mfma_f32_16x16x16_f16 alone", w);
        run<160, 8>("160 fma + 8 mfma_f32_16x16x16_f16", w);
        run<0, 16>("16 mfma_f32_16x16x16_f16 alone", w);
        run<160, 16>("160 fma + 16 mfma_f32_16x16x16_f16", w);
    }
    return 0;
}
