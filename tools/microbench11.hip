// microbench11.hip -- follow-up to microbench10: v_exp_f32 mixed into a v_fma_f32 stream costs far more than the sum of the parts
// (8 exp + 88 fma: 357 cycles per group per SIMD at 2 waves, against 67 + 211).  What shrinks that?  Wait states after each
// exp (s_nop N), the position of the exps, a plain VOP1 instead of the transcendental (control).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 8192;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define G(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0)
#define ALLF G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7)
#define CLOB "s20","s21","v90","v91","v92","v93","v94","v95","v96","v97", \
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
#define EXN(a, N) "v_exp_f32 v" #a ", v81\n s_nop " #N "\n"
#define BURSTN(N) EXN(90, N) EXN(91, N) EXN(92, N) EXN(93, N) EXN(94, N) EXN(95, N) EXN(96, N) EXN(97, N) ALLF
#define MIXN(N) EXN(90, N) G(0) EXN(91, N) G(1) EXN(92, N) G(2) EXN(93, N) G(3) EXN(94, N) G(4) EXN(95, N) G(5) EXN(96, N) G(6) EXN(97, N) G(7)
#define EX0(a) "v_exp_f32 v" #a ", v81\n"
#define MV(a) "v_mov_b32 v" #a ", v81\n"
PROBE(k_b0, EX0(90) EX0(91) EX0(92) EX0(93) EX0(94) EX0(95) EX0(96) EX0(97) ALLF)
PROBE(k_b1, BURSTN(1))
PROBE(k_b3, BURSTN(3))
PROBE(k_b7, BURSTN(7))
PROBE(k_b15, BURSTN(15))
PROBE(k_m0, EX0(90) G(0) EX0(91) G(1) EX0(92) G(2) EX0(93) G(3) EX0(94) G(4) EX0(95) G(5) EX0(96) G(6) EX0(97) G(7))
PROBE(k_m3, MIXN(3))
PROBE(k_m7, MIXN(7))
PROBE(k_m15, MIXN(15))
PROBE(k_tail, EX0(90) EX0(91) EX0(92) EX0(93) EX0(94) EX0(95) EX0(96) EX0(97) "s_nop 15\n" ALLF)
PROBE(k_mov, MV(90) MV(91) MV(92) MV(93) MV(94) MV(95) MV(96) MV(97) ALLF)
PROBE(k_b0n, BURSTN(0))
PROBE(k_b2, BURSTN(2))
PROBE(k_m0n, MIXN(0))
PROBE(k_m1, MIXN(1))
PROBE(k_m2, MIXN(2))
#define EXS(a) "v_exp_f32 v" #a ", v81\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n"
PROBE(k_bs, EXS(90) EXS(91) EXS(92) EXS(93) EXS(94) EXS(95) EXS(96) EXS(97) ALLF)
#define EXF(a, f) "v_exp_f32 v" #a ", v81\n" F(f)
PROBE(k_alt, EXF(90, 100) EXF(91, 101) EXF(92, 102) EXF(93, 103) EXF(94, 104) EXF(95, 105) EXF(96, 106) EXF(97, 107) G(1) G(2) G(3) G(4) G(5) G(6) G(7) F(108) F(109) F(100))
PROBE(k_bp, "s_setprio 1\n" EX0(90) EX0(91) EX0(92) EX0(93) EX0(94) EX0(95) EX0(96) EX0(97) "s_setprio 0\n" ALLF)
PROBE(k_bpn, "s_setprio 1\n" EXN(90, 0) EXN(91, 0) EXN(92, 0) EXN(93, 0) EXN(94, 0) EXN(95, 0) EXN(96, 0) EXN(97, 0) "s_setprio 0\n" ALLF)
PROBE(k_bpf, EX0(90) EX0(91) EX0(92) EX0(93) EX0(94) EX0(95) EX0(96) EX0(97) "s_setprio 1\n" ALLF "s_setprio 0\n")
PROBE(k_f, ALLF)

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
        printf("%-46s waves/SIMD=%d  %.1f cycles per group per SIMD (wall %.3f ms at %.0f MHz)\n", name, wps, ms * 1e-3 * clk * 1e6 / ((double)ITERS * wps), ms, clk);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
int main()
{
    run("88 fma", k_f); run("8 v_mov (control) + 88 fma", k_mov);
    run("burst 8 exp + 88 fma", k_b0); run("burst, s_nop 1 after each exp", k_b1); run("burst, s_nop 3 after each exp", k_b3);
    run("burst, s_nop 7 after each exp", k_b7); run("burst, s_nop 15 after each exp", k_b15); run("burst, one s_nop 15 after the 8 exps", k_tail);
    run("interleaved (1 exp, 11 fma) x 8", k_m0); run("interleaved, s_nop 3 after each exp", k_m3); run("interleaved, s_nop 7 after each exp", k_m7);
    run("interleaved, s_nop 15 after each exp", k_m15);
    run("burst, s_nop 0 after each exp", k_b0n); run("burst, s_nop 2 after each exp", k_b2);
    run("interleaved, s_nop 0 after each exp", k_m0n); run("interleaved, s_nop 1 after each exp", k_m1); run("interleaved, s_nop 2 after each exp", k_m2);
    run("burst, s_setprio 1 around the 8 exps", k_bp); run("burst, s_setprio 1 around (exp + s_nop 0) x 8", k_bpn);
    run("burst, s_setprio 1 around the 88 fma", k_bpf);
    run("burst, two s_mov after each exp", k_bs); run("exp,fma alternating x8, then 80 fma", k_alt);
    return 0;
}
