// microbench.hip -- gfx950 instruction-throughput probes used to size the NLM / bilateral
// kernels (development aid, not part of the library).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    __shared__ float4 sm[1024];
    if (KIND >= 5) { for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = make_float4(seed, 1.f, 2.f, 3.f); __syncthreads(); }
    const float b = seed * 0.5f, c = seed * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) {   // v_fma_f32 x16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 1) {   // v_pk_fma_f32 x8 (16 fmas)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {a[i], a[i + 1]}, bb = {b, b}, cc = {c, c};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(cc));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (KIND == 2) {   // v_exp_f32 x16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 3) {   // v_add_f32 dpp wave_shr:1 x16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 4) {   // v_add_f32 dpp row_shr:1 x16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 5) {   // ds_read_b128 x8
            const float4 *p = sm + (threadIdx.x & 63) + ((it & 7) << 6);
#pragma unroll
            for (int i = 0; i < 8; ++i) { float4 v = p[(i & 3) * 128]; asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); a[i] += v.x; }
        } else if (KIND == 6) {   // ds_read_b96 x8
            const float4 *p = sm + (threadIdx.x & 63) + ((it & 7) << 6);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                typedef float f3 __attribute__((ext_vector_type(3)));
                f3 v = *(const f3 *)(p + (i & 3) * 128); asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z)); a[i] += v.x; }
        } else if (KIND == 7) {   // v_pk_add_f32 x8
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {a[i], a[i + 1]}, bb = {b, b};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(bb));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (KIND == 8) {   // v_sub + v_mul + v_fma mix like the D computation (6 ops) x2 + exp
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i + 1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i + 2]) : "v"(b), "v"(c));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i + 3]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(const char *name, double ops_per_iter_per_lane, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves = 1 wave/SIMD each
    float *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = (double)blocks * 256 * ITERS * ops_per_iter_per_lane;
    const double wave_instr_per_simd = (double)waves_per_simd * ITERS * ops_per_iter_per_lane;  // per SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Tlane-op/s  %.2f clk/wave-instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
           lane_ops / ms / 1e9, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
    CK(hipFree(d));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 16, w);
        run<1>("v_pk_fma_f32 (per pk instr)", 8, w);
        run<7>("v_pk_add_f32 (per pk instr)", 8, w);
        run<2>("v_exp_f32", 16, w);
        run<3>("v_add_f32_dpp wave_shr:1", 16, w);
        run<4>("v_add_f32_dpp row_shr:1", 16, w);
        run<5>("ds_read_b128", 8, w);
        run<6>("ds_read_b96", 8, w);
        run<8>("mix sub/mul/fma/exp", 16, w);
    }
    return 0;
}
