// microbench16.hip -- the horizontal box sums by TRANSPOSING through LDS memory instead of DPP, best case (conflict-free addresses):
//   dpp   : 48 v_add_f32_dpp + 8 v_exp_f32 at s_setprio 1, then 144 v_fma_f32                         (the kernel's offset, without its tile reads)
//   trans : 2 ds_write_b128 (the 8 vertical sums of a lane) + 4 ds_read_b128 (14 consecutive columns of one row) + wait,
//           18 v_add_f32 (sliding sums in registers) + 8 v_exp_f32 at s_setprio 1,
//           2 ds_write_b128 (8 weights) + 2 ds_read_b128 (back in the column-per-lane layout) + wait, then 144 v_fma_f32
// A wave's LDS operations execute in order, so no barrier sits between its writes and its reads.  Two 4-wave workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define D(a) "v_add_f32_dpp v" #a ", v" #a ", v80 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A(a) "v_add_f32 v" #a ", v" #a ", v80\n"
#define E(a) "v_exp_f32 v" #a ", v" #a "\n"
#define F12(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0) F(1##b##1)
#define ALLF F12(0) F12(1) F12(2) F12(3) F12(4) F12(5) F12(6) F12(7) F12(0) F12(1) F12(2) F12(3)
#define D4(b) D(2##b##0) D(2##b##1) D(2##b##2) D(2##b##3)
#define ALLD D4(0) D4(1) D4(2) D4(3) D4(4) D4(5) D4(0) D4(1) D4(2) D4(3) D4(4) D4(5)
#define E8 E(200) E(201) E(202) E(203) E(210) E(211) E(212) E(213)
#define A18 A(220) A(221) A(222) A(223) A(230) A(231) A(232) A(233) A(240) A(241) A(242) A(243) A(250) A(251) A(252) A(253) A(220) A(221)
#define WR(r, off) "ds_write_b128 v82, v[" #r ":" #r "+3] offset:" #off "\n"
#define RD(r, off) "ds_read_b128 v[" #r ":" #r "+3], v82 offset:" #off "\n"
#define W0 "s_waitcnt lgkmcnt(0)\n"
#define HI "s_setprio 1\n"
#define LO "s_setprio 0\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179", \
  "v200","v201","v202","v203","v210","v211","v212","v213","v220","v221","v222","v223","v230","v231","v232","v233","v240","v241","v242","v243","v250","v251","v252","v253", \
  "v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199"
#define PROBE(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        __shared__ float4 scratch[4 * 128];   /* 2 KB per wave */                                          \
        scratch[threadIdx.x] = make_float4(seed, seed, seed, seed); scratch[threadIdx.x + 256] = make_float4(seed, seed, seed, seed); \
        __syncthreads();                                                                                   \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %1\n" :: "v"(seed * 1e-3f),  \
                     "v"((unsigned)(((threadIdx.x >> 6) * 128 + (threadIdx.x & 63)) * 16)) : "v80", "v81", "v82"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) asm volatile(BODY ::: CLOB, "memory");                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v200\n v_add_f32 %0, %0, v177\n v_add_f32 %0, %0, v193" : "=v"(s));  \
        out[blockIdx.x * 256 + threadIdx.x] = s + scratch[threadIdx.x].x;                                  \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
// first transposition: the 8 vertical sums (v200..v203, v210..v213 stand for them) out, 16 floats of one row in (v184..v199)
#define T1 WR(200, 0) WR(210, 1024) RD(184, 0) RD(188, 1024) RD(192, 0) RD(196, 1024) W0
// second: 8 weights out, 8 back
#define T2 WR(220, 0) WR(230, 1024) RD(240, 0) RD(250, 1024) W0
PROBE(k_dpp, HI ALLD E8 LO ALLF)
PROBE(k_trans, T1 HI A18 E8 LO T2 ALLF)
PROBE(k_trans_hi, HI T1 A18 E8 T2 LO ALLF)
PROBE(k_trans_early, WR(200, 0) WR(210, 1024) RD(184, 0) RD(188, 1024) RD(192, 0) RD(196, 1024) F12(0) F12(1) F12(2) F12(3) W0 HI A18 E8 LO WR(220, 0) WR(230, 1024) RD(240, 0) RD(250, 1024) F12(4) F12(5) F12(6) F12(7) W0 F12(0) F12(1) F12(2) F12(3))
PROBE(k_floor, HI A18 E8 LO ALLF)
PROBE(k_lds_only, T1 T2)

template <typename K>
int run(const char *name, K kern)
{
    const int wps = 2, blocks = 256 * wps;
    float *d; unsigned long long *c;
    CK(hipMalloc(&d, (size_t)blocks * 256 * 4)); CK(hipMalloc(&c, (size_t)blocks * 4 * 16));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, c, 1.0f);
    CK(hipDeviceSynchronize());
    double best = 1e30, clkb = 0;
    for (int rep = 0; rep < 3; ++rep) {
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
        const double clk = mhz[mhz.size() / 2], cyc = ms * 1e-3 * clk * 1e6 / ((double)ITERS * wps);
        if (cyc < best) { best = cyc; clkb = clk; }
    }
    printf("%-72s %.1f cycles per group per SIMD (2 waves/SIMD, best of 3, %.0f MHz)\n", name, best, clkb);
    CK(hipFree(d)); CK(hipFree(c));
    return 0;
}
int main()
{
    run("48 dpp + 8 exp at raised priority, 144 fma   [the kernel's offset]", k_dpp);
    run("18 add + 8 exp at raised priority, 144 fma   [no cross-lane work at all: the floor]", k_floor);
    run("the two transpositions alone (4 ds_write_b128 + 6 ds_read_b128 + 2 waits)", k_lds_only);
    run("transpose, 18 add + 8 exp raised, transpose back, 144 fma", k_trans);
    run("the same, priority raised over the transpositions too", k_trans_hi);
    run("the same, 48 fma between each transposition's issue and its wait", k_trans_early);
    return 0;
}
