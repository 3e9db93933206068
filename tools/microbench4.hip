// microbench4.hip -- are some DPP modes cheaper than others on gfx950?  (v_add_f32 with each control,
// 16 independent accumulators, timed beside nothing else.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define PROBE(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed)                                    \
    {                                                                                                      \
        float a[16];                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);            \
        const float b = seed * 0.5f;                                                                       \
        for (int it = 0; it < ITERS; ++it) {                                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b));       \
        }                                                                                                  \
        float s = 0.f;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                                          \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
    }
PROBE(k_plain, "v_add_f32 %0, %0, %1")
PROBE(k_quad, "v_add_f32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")
PROBE(k_rowshr1, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(k_rowshr4, "v_add_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf")
PROBE(k_rowror, "v_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf")
PROBE(k_waveshr, "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(k_waveror, "v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf")
PROBE(k_mirror, "v_add_f32_dpp %0, %0, %1 row_mirror row_mask:0xf bank_mask:0xf")
PROBE(k_movdpp, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(k_sdwa, "v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")

template <typename K>
int run(const char *name, K kern)
{
    for (int wps : {2, 8}) {
        const int blocks = 256 * wps;
        float *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s waves/SIMD=%d  %.3f ms  %.2f clk(2.4GHz)/instr/SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)wps * ITERS * 16));
        CK(hipFree(d));
    }
    return 0;
}

int main()
{
    run("v_add_f32", k_plain); run("dpp quad_perm", k_quad); run("dpp row_shr:1", k_rowshr1); run("dpp row_shr:4", k_rowshr4);
    run("dpp row_ror:1", k_rowror); run("dpp wave_shr:1", k_waveshr); run("dpp wave_ror:1", k_waveror); run("dpp row_mirror", k_mirror);
    run("v_mov_b32_dpp wave_shr", k_movdpp); run("v_add_f32_sdwa", k_sdwa);
    return 0;
}
