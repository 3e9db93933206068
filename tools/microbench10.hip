// microbench10.hip -- does the transcendental pipe (v_exp_f32, quarter rate) run BESIDE the plain vector ALU on gfx950?
// 8 v_exp_f32 + 88 v_fma_f32 per group, once as a burst (8 exps, then 88 FMAs: how the NLM loop issues them) and once interleaved
// (1 exp, 11 FMAs, ...), against the two alone.  If the pipes overlap, a group costs less than the sum of its parts.
// Cycles from s_memtime/s_memrealtime as in microbench8; 2 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 8192;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define E(a) "v_exp_f32 v" #a ", v" #a "\n"
#define F11(b) F(b) F(b##1) F(b##2) F(b##3) F(b##4) F(b##5) F(b##6) F(b##7) F(b##8) F(b##9) F(b)
// FMA accumulators v100..v109, v110..v119 ...: use 8 groups of 11 over registers 100..187 (88 distinct)
#define G(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0)
#define ALLF G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7)
#define ALLE E(90) E(91) E(92) E(93) E(94) E(95) E(96) E(97)
#define MIX E(90) G(0) E(91) G(1) E(92) G(2) E(93) G(3) E(94) G(4) E(95) G(5) E(96) G(6) E(97) G(7)
#define CLOB "v90","v91","v92","v93","v94","v95","v96","v97", \
  "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179"
#define PROBE(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n" :: "v"(seed * 1e-3f) : "v80", "v81");     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) asm volatile(BODY ::: CLOB);                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v90\n v_add_f32 %0, %0, v177" : "=v"(s));                       \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
#define A(a) "v_add_f32 v" #a ", v80, v" #a "\n"
#define GA(b) A(1##b##0) A(1##b##1) A(1##b##2) A(1##b##3) A(1##b##4) A(1##b##5) A(1##b##6) A(1##b##7) A(1##b##8) A(1##b##9) A(1##b##0)
#define ALLA GA(0) GA(1) GA(2) GA(3) GA(4) GA(5) GA(6) GA(7)
#define R(a) "v_rcp_f32 v" #a ", v" #a "\n"
#define ALLR R(90) R(91) R(92) R(93) R(94) R(95) R(96) R(97)
#define EC(a) "v_exp_f32 v" #a ", v81\n"          /* constant source: no chain through the exps */
#define ALLEC EC(90) EC(91) EC(92) EC(93) EC(94) EC(95) EC(96) EC(97)
#define EN(a) "v_exp_f32 v" #a ", v" #a "\n s_nop 3\n"
#define ALLEN EN(90) EN(91) EN(92) EN(93) EN(94) EN(95) EN(96) EN(97)
PROBE(k_add88, ALLA)
PROBE(k_burst_add, ALLE ALLA)
PROBE(k_burst_rcp, ALLR ALLF)
PROBE(k_burst_const, ALLEC ALLF)
PROBE(k_burst_nop, ALLEN ALLF)
PROBE(k_burst16, ALLE ALLE ALLF)
PROBE(k_burst_176, ALLE ALLF ALLF)
PROBE(k_fma88, ALLF)
PROBE(k_exp8, ALLE)
PROBE(k_burst, ALLE ALLF)
PROBE(k_mix, MIX)

template <typename K>
int run(const char *name, K kern, int per_group)
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
        printf("%-44s waves/SIMD=%d  %.1f cycles per group per SIMD (%d instructions; wall %.3f ms at %.0f MHz)\n", name, wps,
               ms * 1e-3 * clk * 1e6 / ((double)ITERS * wps), per_group, ms, clk);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
int main()
{
    run("88 v_fma_f32", k_fma88, 88); run("8 v_exp_f32", k_exp8, 8);
    run("burst: 8 exp, then 88 fma", k_burst, 96); run("interleaved: (1 exp, 11 fma) x 8", k_mix, 96);
    run("88 v_add_f32", k_add88, 88); run("burst: 8 exp, then 88 add", k_burst_add, 96);
    run("burst: 8 rcp, then 88 fma", k_burst_rcp, 96); run("burst: 8 exp (constant source), 88 fma", k_burst_const, 96);
    run("burst: 8 (exp + s_nop 3), 88 fma", k_burst_nop, 96); run("burst: 16 exp, then 88 fma", k_burst16, 104);
    run("burst: 8 exp, then 176 fma", k_burst_176, 184);
    return 0;
}
