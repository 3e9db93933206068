// microbench14.hip -- can the LDS crossbar take the horizontal box sums off the vector pipe?
// The NLM loop's 7-wide horizontal sum costs 6 whole-wave DPP adds per row (4.4 cycles each).  ds_bpermute_b32 moves a
// register between arbitrary lanes through the LDS unit (no memory, no VALU slot) and a plain add costs 2.2 cycles, so
// one DPP add (pairs) + 4 permutes + 3 plain adds would do the same sum.  Per group of 8 rows, beside the loop's 144 FMAs:
//   dpp   : 48 v_add_f32_dpp                                 (what the kernel does)
//   perm  :  8 v_add_f32_dpp + 32 ds_bpermute_b32 + 24 v_add_f32
// plus the permute stream alone, for the LDS unit's rate at 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define D(a) "v_add_f32_dpp v" #a ", v" #a ", v80 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A(a) "v_add_f32 v" #a ", v" #a ", v80\n"
#define B(d, s) "ds_bpermute_b32 v" #d ", v82, v" #s "\n"
#define F12(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0) F(1##b##1)
#define ALLF F12(0) F12(1) F12(2) F12(3) F12(4) F12(5) F12(6) F12(7) F12(0) F12(1) F12(2) F12(3)
#define D4(b) D(2##b##0) D(2##b##1) D(2##b##2) D(2##b##3)
#define ALLD D4(0) D4(1) D4(2) D4(3) D4(4) D4(5) D4(0) D4(1) D4(2) D4(3) D4(4) D4(5)
#define D8 D(203) D(213) D(223) D(233) D(243) D(253) D(183) D(193)
#define B4(b) B(2##b##0, 1##b##0) B(2##b##1, 1##b##1) B(2##b##2, 1##b##2) B(2##b##3, 1##b##3)
#define B4X B(180, 160) B(181, 161) B(182, 162) B(183, 163)
#define B4Y B(190, 170) B(191, 171) B(192, 172) B(193, 173)
#define ALLB B4(0) B4(1) B4(2) B4(3) B4(4) B4(5) B4X B4Y
#define HALFB0 B4(0) B4(1) B4(2) B4(3)
#define HALFB1 B4(4) B4(5) B4X B4Y
#define A3(b) A(2##b##0) A(2##b##1) A(2##b##2)
#define A3X A(180) A(181) A(182)
#define A3Y A(190) A(191) A(192)
#define ALLA A3(0) A3(1) A3(2) A3(3) A3(4) A3(5) A3X A3Y
#define HALFA0 A3(0) A3(1) A3(2) A3(3)
#define HALFA1 A3(4) A3(5) A3X A3Y
#define W0 "s_waitcnt lgkmcnt(0)\n"
#define W16 "s_waitcnt lgkmcnt(15)\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179", \
  "v200","v201","v202","v203","v210","v211","v212","v213","v220","v221","v222","v223","v230","v231","v232","v233","v240","v241","v242","v243","v250","v251","v252","v253", \
  "v180","v181","v182","v183","v190","v191","v192","v193"
#define PROBE(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %1\n" :: "v"(seed * 1e-3f), "v"(((threadIdx.x + 3) & 63) * 4) : "v80", "v81", "v82"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) asm volatile(BODY ::: CLOB);                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v200\n v_add_f32 %0, %0, v177\n v_add_f32 %0, %0, v193" : "=v"(s));  \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
PROBE(k_f, ALLF)
PROBE(k_b, ALLB W0)
PROBE(k_dpp, ALLD ALLF)
PROBE(k_dpp_hi, "s_setprio 1\n" ALLD "s_setprio 0\n" ALLF)
PROBE(k_perm, D8 ALLB W0 ALLA ALLF)
PROBE(k_perm_hi, "s_setprio 1\n" D8 ALLB W0 ALLA "s_setprio 0\n" ALLF)
PROBE(k_perm_hi2, "s_setprio 1\n" D8 HALFB0 HALFB1 W16 HALFA0 W0 HALFA1 "s_setprio 0\n" ALLF)
PROBE(k_perm_lo, D8 ALLB "s_setprio 0\n" W0 "s_setprio 1\n" ALLA ALLF)
PROBE(k_perm_early, D8 ALLB F12(0) F12(1) F12(2) F12(3) W0 ALLA F12(4) F12(5) F12(6) F12(7) F12(0) F12(1) F12(2) F12(3))
PROBE(k_perm_only_adds, D8 ALLA ALLF)

template <typename K>
int run(const char *name, K kern)
{
    for (int wps : {1, 2}) {
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
        printf("%-60s waves/SIMD=%d  %.1f cycles per group per SIMD (wall %.3f ms at %.0f MHz)\n", name, wps, ms * 1e-3 * clk * 1e6 / ((double)ITERS * wps), ms, clk);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
int main()
{
    run("144 v_fma_f32", k_f);
    run("32 ds_bpermute_b32 + wait", k_b);
    run("48 dpp, then 144 fma", k_dpp);
    run("48 dpp at s_setprio 1, then 144 fma", k_dpp_hi);
    run("8 dpp + 24 add + 144 fma (no permutes: the VALU floor)", k_perm_only_adds);
    run("8 dpp + 32 bpermute + wait + 24 add, then 144 fma", k_perm);
    run("the same, dpp..add at s_setprio 1", k_perm_hi);
    run("the same, waits split 16/16", k_perm_hi2);
    run("the same, priority dropped while waiting", k_perm_lo);
    run("permutes issued 48 fma before their wait", k_perm_early);
    return 0;
}
