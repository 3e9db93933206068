// microbench15.hip -- the 7-wide horizontal box sum of x of the NLM loop's 8 rows through LDS MEMORY instead of DPP:
//   DPP row : 6 v_add_f32_dpp                                                     (26 VALU cycles, no LDS)
//   LDS row : a = v + shl1(v) (1 DPP add); ds_write2_b32 {a, v}; 2 ds_read2_b32 {a[l-3], a[l-1]}, {a[l+1], v[l+3]}; 3 v_add_f32
//             (11 VALU cycles + 14 LDS cycles; a wave's LDS operations execute in order, so no wait between its write and reads)
// The LDS rows are issued first, the DPP rows cover their latency, one wait, then the plain adds; 144 FMAs stand for the rest of
// the loop; the whole sum phase runs at s_setprio 1 as in the kernel.  2 workgroups of 4 waves per CU = the kernel's occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define D(a) "v_add_f32_dpp v" #a ", v" #a ", v80 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define F12(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0) F(1##b##1)
#define ALLF F12(0) F12(1) F12(2) F12(3) F12(4) F12(5) F12(6) F12(7) F12(0) F12(1) F12(2) F12(3)
#define R6 D(180) D(181) D(182) D(183) D(190) D(191)
#define L(b) D(1##b##1) "ds_write2_b32 v82, v1" #b "1, v1" #b "0 offset0:3 offset1:75\n" \
             "ds_read2_b32 v[2" #b "0:2" #b "1], v82 offset1:2\n" "ds_read2_b32 v[2" #b "2:2" #b "3], v82 offset0:4 offset1:78\n"
#define L1(b) D(1##b##1) "ds_write_b32 v82, v1" #b "1 offset:12\n ds_write_b32 v82, v1" #b "0 offset:300\n" \
             "ds_read2_b32 v[2" #b "0:2" #b "1], v82 offset1:2\n" "ds_read2_b32 v[2" #b "2:2" #b "3], v82 offset0:4 offset1:78\n"
#define LA(b) "v_add_f32 v2" #b "0, v2" #b "0, v2" #b "1\n v_add_f32 v2" #b "2, v2" #b "2, v2" #b "3\n v_add_f32 v2" #b "0, v2" #b "0, v2" #b "2\n"
#define W0 "s_waitcnt lgkmcnt(0)\n"
#define HI "s_setprio 1\n"
#define LO "s_setprio 0\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179", \
  "v200","v201","v202","v203","v210","v211","v212","v213","v220","v221","v222","v223","v230","v231","v232","v233","v240","v241","v242","v243","v250","v251","v252","v253", \
  "v180","v181","v182","v183","v190","v191","v192","v193"
#define PROBE(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        __shared__ float scratch[4 * 160];                                                                 \
        scratch[threadIdx.x] = seed; scratch[threadIdx.x + 256] = seed; if (threadIdx.x < 128) scratch[threadIdx.x + 512] = seed; \
        __syncthreads();                                                                                   \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %1\n" :: "v"(seed * 1e-3f),  \
                     "v"((unsigned)(((threadIdx.x >> 6) * 160 + (threadIdx.x & 63)) * 4)) : "v80", "v81", "v82"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) asm volatile(BODY ::: CLOB, "memory");                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v200\n v_add_f32 %0, %0, v177\n v_add_f32 %0, %0, v193" : "=v"(s));  \
        out[blockIdx.x * 256 + threadIdx.x] = s + scratch[threadIdx.x];                                    \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
PROBE(k_x0, HI R6 R6 R6 R6 R6 R6 R6 R6 LO ALLF)
PROBE(k_x2, HI L(0) L(1) R6 R6 R6 R6 R6 R6 W0 LA(0) LA(1) LO ALLF)
PROBE(k_x3, HI L(0) L(1) L(2) R6 R6 R6 R6 R6 W0 LA(0) LA(1) LA(2) LO ALLF)
PROBE(k_x4, HI L(0) L(1) L(2) L(3) R6 R6 R6 R6 W0 LA(0) LA(1) LA(2) LA(3) LO ALLF)
PROBE(k_x5, HI L(0) L(1) L(2) L(3) L(4) R6 R6 R6 W0 LA(0) LA(1) LA(2) LA(3) LA(4) LO ALLF)
PROBE(k_x6, HI L(0) L(1) L(2) L(3) L(4) L(5) R6 R6 W0 LA(0) LA(1) LA(2) LA(3) LA(4) LA(5) LO ALLF)
PROBE(k_x4_w1, HI L1(0) L1(1) L1(2) L1(3) R6 R6 R6 R6 W0 LA(0) LA(1) LA(2) LA(3) LO ALLF)
PROBE(k_x4_noprio, L(0) L(1) L(2) L(3) R6 R6 R6 R6 W0 LA(0) LA(1) LA(2) LA(3) ALLF)
PROBE(k_x4_late, HI R6 R6 R6 R6 L(0) L(1) L(2) L(3) W0 LA(0) LA(1) LA(2) LA(3) LO ALLF)
PROBE(k_x4_lowait, HI L(0) L(1) L(2) L(3) R6 R6 R6 R6 LO W0 HI LA(0) LA(1) LA(2) LA(3) LO ALLF)

template <typename K>
int run(const char *name, K kern)
{
    for (int wps : {2}) {
        const int blocks = 256 * wps;
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
        printf("%-64s waves/SIMD=%d  %.1f cycles per group per SIMD (best of 3, %.0f MHz)\n", name, wps, best, clkb);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
int main()
{
    run("8 DPP rows (48 dpp) + 144 fma   [the kernel's phase]", k_x0);
    run("2 LDS rows + 6 DPP rows", k_x2);
    run("3 LDS rows + 5 DPP rows", k_x3);
    run("4 LDS rows + 4 DPP rows", k_x4);
    run("5 LDS rows + 3 DPP rows", k_x5);
    run("6 LDS rows + 2 DPP rows", k_x6);
    run("4 LDS rows, two ds_write_b32 instead of ds_write2_b32", k_x4_w1);
    run("4 LDS rows, no s_setprio", k_x4_noprio);
    run("4 LDS rows issued AFTER the DPP rows", k_x4_late);
    run("4 LDS rows, priority dropped during the wait", k_x4_lowait);
    return 0;
}
