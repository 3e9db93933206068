// microbench12.hip -- the same question as microbench10 for DPP: is a whole-wave DPP add beside plain VALU work additive?
// 48 v_add_f32_dpp (wave_shl:1) + 144 v_fma_f32 per group (the NLM loop's proportions), burst vs interleaved vs wait states.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define D(a) "v_add_f32_dpp v" #a ", v" #a ", v80 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DN(a) D(a) "s_nop 0\n"
#define F3(b) F(1##b##0) F(1##b##1) F(1##b##2)
#define F12(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0) F(1##b##1)
#define ALLF F12(0) F12(1) F12(2) F12(3) F12(4) F12(5) F12(6) F12(7) F12(0) F12(1) F12(2) F12(3)
#define D4(b) D(2##b##0) D(2##b##1) D(2##b##2) D(2##b##3)
#define ALLD D4(0) D4(1) D4(2) D4(3) D4(4) D4(5) D4(0) D4(1) D4(2) D4(3) D4(4) D4(5)
#define DN4(b) DN(2##b##0) DN(2##b##1) DN(2##b##2) DN(2##b##3)
#define ALLDN DN4(0) DN4(1) DN4(2) DN4(3) DN4(4) DN4(5) DN4(0) DN4(1) DN4(2) DN4(3) DN4(4) DN4(5)
#define MIX1(b, c) D4(b) F12(c)
#define MIX MIX1(0,0) MIX1(1,1) MIX1(2,2) MIX1(3,3) MIX1(4,4) MIX1(5,5) MIX1(0,6) MIX1(1,7) MIX1(2,0) MIX1(3,1) MIX1(4,2) MIX1(5,3)
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179", \
  "v200","v201","v202","v203","v210","v211","v212","v213","v220","v221","v222","v223","v230","v231","v232","v233","v240","v241","v242","v243","v250","v251","v252","v253"
#define PROBE(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n" :: "v"(seed * 1e-3f) : "v80", "v81");     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) asm volatile(BODY ::: CLOB);                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v200\n v_add_f32 %0, %0, v177" : "=v"(s));                      \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
#define PROBE8(NAME, BODY, BAR)                                                                            \
    __global__ __launch_bounds__(512) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n" :: "v"(seed * 1e-3f) : "v80", "v81");     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) { if (BAR) __builtin_amdgcn_s_barrier(); asm volatile(BODY ::: CLOB); } \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v200\n v_add_f32 %0, %0, v177" : "=v"(s));                      \
        out[blockIdx.x * 512 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
PROBE8(k8_burst, ALLD ALLF, 0)
PROBE8(k8_burst_bar, ALLD ALLF, 1)
PROBE8(k8_f_bar, ALLF, 1)
PROBE(k_burst_hi, "s_setprio 3\n" ALLD "s_setprio 0\n" ALLF)
PROBE(k_burst_lo, "s_setprio 0\n" ALLD "s_setprio 3\n" ALLF)
PROBE(k_burst_hi1, "s_setprio 1\n" ALLD "s_setprio 0\n" ALLF)
PROBE(k_f, ALLF)
PROBE(k_d, ALLD)
PROBE(k_burst, ALLD ALLF)
PROBE(k_burst_n, ALLDN ALLF)
PROBE(k_mix, MIX)

template <typename K>
int run(const char *name, K kern)
{
    for (int wps : {2, 8}) {
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
        std::vector<double> mhz;
        for (size_t i = 0; i < h.size(); i += 2) mhz.push_back((double)h[i] / (double)h[i + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double clk = mhz[mhz.size() / 2];
        printf("%-44s waves/SIMD=%d  %.1f cycles per group per SIMD (wall %.3f ms at %.0f MHz)\n", name, wps, ms * 1e-3 * clk * 1e6 / ((double)ITERS * wps), ms, clk);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
template <typename K>
int run8(const char *name, K kern)
{
    const int blocks = 256;                       // one 8-wave workgroup per CU: 2 waves per SIMD from the SAME workgroup
    float *d; unsigned long long *c;
    CK(hipMalloc(&d, (size_t)blocks * 512 * 4)); CK(hipMalloc(&c, (size_t)blocks * 8 * 16));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, c, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, c, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 8 * 2);
    CK(hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (size_t i = 0; i < h.size(); i += 2) mhz.push_back((double)h[i] / (double)h[i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double clk = mhz[mhz.size() / 2];
    printf("%-44s waves/SIMD=2  %.1f cycles per group per SIMD (wall %.3f ms at %.0f MHz)\n", name, ms * 1e-3 * clk * 1e6 / ((double)ITERS * 2), ms, clk);
    return 0;
}
int main()
{
    run8("8-wave WG: burst 48 dpp + 144 fma", k8_burst); run8("8-wave WG: the same, s_barrier per group", k8_burst_bar);
    run8("8-wave WG: 144 fma, s_barrier per group", k8_f_bar);
    run("144 v_fma_f32", k_f); run("48 v_add_f32_dpp", k_d); run("burst: 48 dpp, then 144 fma", k_burst);
    run("burst, s_nop 0 after each dpp", k_burst_n);
    run("burst, s_setprio 3 during the dpp phase", k_burst_hi); run("burst, s_setprio 1 during the dpp phase", k_burst_hi1);
    run("burst, s_setprio 3 during the fma phase", k_burst_lo); run("interleaved: (4 dpp, 12 fma) x 12", k_mix);
    return 0;
}
