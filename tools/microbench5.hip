// microbench5.hip -- dependent-issue latency on gfx950: 16 VALU instructions per iteration arranged as ILP
// independent chains (ILP=1: one 16-deep chain ... ILP=16: all independent), for plain adds, FMAs, whole-wave
// DPP adds (the chain variable is the DPP source, as in the NLM horizontal box sum) and exp.  With 1 and 2
// waves per SIMD: is a 6-deep chain of adds exposed at the occupancy the NLM kernel runs at?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench5.hip -o /tmp/mb5 && /tmp/mb5
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

template <int OP, int ILP>
__global__ __launch_bounds__(256) void probe(float *out, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const float b = seed * 0.5f, c = seed * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float &x = a[i % ILP];
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 2) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            if (OP == 4) {   // the NLM tail: dpp add -> exp -> fma, per chain step (3 instructions)
                if (i % 3 == 0) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
                if (i % 3 == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (i % 3 == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int ILP>
int run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    float *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_wave = ms * 1e-3 * 2.4e9 / ((double)ITERS * 16);            // clk per instruction as one wave sees it
    printf("%-14s ILP=%-2d waves/SIMD=%d  %.3f ms  %.2f clk/instr per wave  %.2f clk/instr per SIMD (@2.4GHz)\n", name, ILP,
           waves_per_simd, ms, per_wave, per_wave / waves_per_simd);
    CK(hipFree(d));
    return 0;
}

template <int OP>
int sweep(const char *name)
{
    for (int w : {1, 2, 4}) {
        if (run<OP, 1>(name, w) || run<OP, 2>(name, w) || run<OP, 4>(name, w) || run<OP, 8>(name, w) || run<OP, 16>(name, w)) return 1;
    }
    return 0;
}

int main()
{
    return sweep<0>("v_add_f32") || sweep<1>("v_fma_f32") || sweep<2>("v_add_dpp") || sweep<3>("v_exp_f32") || sweep<4>("dpp-exp-fma");
}
