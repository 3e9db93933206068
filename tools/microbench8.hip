// microbench8.hip -- the issue floor of plain fp32 VALU instructions on gfx950 in SHADER CYCLES, not in an assumed clock.
// Each wave runs a long unrolled stream (128 independent-enough instructions per loop trip, ~1 M per wave) and brackets it with
// s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz): the ratio gives the clock the chip actually held, the
// s_memtime delta / (instructions per wave x waves per SIMD) the cycles per wave-instruction per SIMD.  Grids are sized so that
// every wave is resident from start to end (256 CUs x `wps` workgroups of 4 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 8192, NACC = 16, REP = 8;

#define PROBE(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        float a[NACC];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < NACC; ++i) a[i] = seed + (float)(threadIdx.x + i);          \
        const float b = seed * 0.5f + (float)threadIdx.x * 1e-9f, c = seed * 0.25f;                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                                               \
            _Pragma("unroll") for (int r = 0; r < REP; ++r)                                                \
                _Pragma("unroll") for (int i = 0; i < NACC; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s = 0.f;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < NACC; ++i) s += a[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
PROBE(k_add, "v_add_f32 %0, %0, %1")
PROBE(k_mul, "v_mul_f32 %0, %0, %1")
PROBE(k_fma, "v_fma_f32 %0, %1, %2, %0")
PROBE(k_fmac, "v_fmac_f32 %0, %1, %2")
PROBE(k_dpp, "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(k_exp, "v_exp_f32 %0, %0")

template <typename K>
int run(const char *name, K kern)
{
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * wps;
        float *d; unsigned long long *c;
        CK(hipMalloc(&d, (size_t)blocks * 256 * 4)); CK(hipMalloc(&c, (size_t)blocks * 4 * 16));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, c, 1.0f);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, c, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
        CK(hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, mhz;
        for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back((double)h[i]); mhz.push_back((double)h[i] / (double)h[i + 1] * 100.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
        const double n_instr = (double)ITERS * REP * NACC;
        const double med = cyc[cyc.size() / 2], mx = cyc.back();
        printf("%-22s waves/SIMD=%d  %.2f cyc per wave-instr per SIMD (median wave; slowest %.2f)  clock %.0f MHz  | wall %.3f ms -> %.2f cyc at that clock\n",
               name, wps, med / n_instr / wps, mx / n_instr / wps, mhz[mhz.size() / 2], ms, ms * 1e-3 * mhz[mhz.size() / 2] * 1e6 / (n_instr * wps));
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}

int main()
{
    run("v_add_f32 v,v", k_add); run("v_mul_f32 v,v", k_mul); run("v_fma_f32 v,v,v", k_fma); run("v_fmac_f32 v,v", k_fmac);
    run("v_add_f32 dpp wave_shr", k_dpp); run("v_exp_f32", k_exp);
    return 0;
}
