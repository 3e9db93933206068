// microbench18.hip -- microbench17's two-stream probe with MATRIX instructions as one of the streams (round 4): does the matrix pipe
// run beside the other wave's vector instructions?  (microbench3/7 found no overlap with both kinds in ONE stream; the DPP lesson of
// microbench17 is that same-stream tests can mislead.)  Classes added: v_mfma_f32_16x16x16_f16, v_mfma_f32_32x32x16_bf16,
// v_mfma_f32_16x16x4_f32, each on four independent accumulators.
// -- original header of microbench17 follows --
// microbench17.hip -- what the two waves of a SIMD can issue BESIDE each other on gfx950 (round 4).
//
// microbench8/10/12 price instruction classes with both waves of a SIMD running the SAME stream.  The NLM loop at raised
// issue priority does something else: one wave's DPP adds / exps run beside the OTHER wave's plain instructions, and the
// kernel then needs fewer cycles than the same-stream prices add up to (profiles/r03_utilisation.json: "utilisation" 1.03
// against a table that was meant to be a floor).  This probe runs two DIFFERENT streams on the two waves of every SIMD --
// one 8-wave workgroup per CU, waves 0-3 stream A, waves 4-7 stream B (wave w and w+4 land on the same SIMD; the probe
// reads HW_ID and reports how many pairs really did) -- each for a fixed budget of shader cycles (s_memtime), counting
// how many groups of 144 instructions it retires.  From the pair's counts:
//     cycles per SIMD = budget;  price(A beside B) = (budget - nB * solo(B)) / nA   with solo(B) = B's same-stream price.
// A class whose "beside plain" price is below its same-stream price co-issues; the floor table of
// tools/summarize_profiles.py takes, per class, the lowest price any arrangement reaches (never below 2 cycles, the
// SIMD-32 issue time of a wave64 instruction).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"                      /* three VGPR sources */
#define A2(a) "v_add_f32 v" #a ", v80, v" #a "\n"                          /* two VGPR sources */
#define D(a) "v_add_f32_dpp v" #a ", v" #a ", v80 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define E(a) "v_exp_f32 v" #a ", v80\n"
#define R12(M, b) M(1##b##0) M(1##b##1) M(1##b##2) M(1##b##3) M(1##b##4) M(1##b##5) M(1##b##6) M(1##b##7) M(1##b##8) M(1##b##9) M(1##b##0) M(1##b##1)
#define R144(M) R12(M, 0) R12(M, 1) R12(M, 2) R12(M, 3) R12(M, 4) R12(M, 5) R12(M, 6) R12(M, 7) R12(M, 0) R12(M, 1) R12(M, 2) R12(M, 3)
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179"

enum { FMA3 = 0, ADD2 = 1, DPP = 2, EXP = 3, IDLE = 4, MF16 = 5, MBF16 = 6, MF32 = 7 };
constexpr int REPS = 4;
constexpr int group_size(int cls) { return cls >= MF16 ? 32 : 144; }
#define MF16_4 "v_mfma_f32_16x16x16_f16 v[100:103], v[80:81], v[82:83], v[100:103]\n v_mfma_f32_16x16x16_f16 v[104:107], v[80:81], v[82:83], v[104:107]\n" \
               "v_mfma_f32_16x16x16_f16 v[108:111], v[80:81], v[82:83], v[108:111]\n v_mfma_f32_16x16x16_f16 v[112:115], v[80:81], v[82:83], v[112:115]\n"
#define MBF16_4 "v_mfma_f32_32x32x16_bf16 v[100:115], v[80:83], v[84:87], v[100:115]\n v_mfma_f32_32x32x16_bf16 v[116:131], v[80:83], v[84:87], v[116:131]\n" \
                "v_mfma_f32_32x32x16_bf16 v[132:147], v[80:83], v[84:87], v[132:147]\n v_mfma_f32_32x32x16_bf16 v[148:163], v[80:83], v[84:87], v[148:163]\n"
#define MF32_4 "v_mfma_f32_16x16x4_f32 v[100:103], v80, v81, v[100:103]\n v_mfma_f32_16x16x4_f32 v[104:107], v80, v81, v[104:107]\n" \
               "v_mfma_f32_16x16x4_f32 v[108:111], v80, v81, v[108:111]\n v_mfma_f32_16x16x4_f32 v[112:115], v80, v81, v[112:115]\n"
#define X8(S) S S S S S S S S

template <int CLS>
__device__ __forceinline__ void group()
{
    if constexpr (CLS == FMA3) asm volatile(R144(F) ::: CLOB);
    if constexpr (CLS == ADD2) asm volatile(R144(A2) ::: CLOB);
    if constexpr (CLS == DPP) asm volatile(R144(D) ::: CLOB);
    if constexpr (CLS == EXP) asm volatile(R144(E) ::: CLOB);
    if constexpr (CLS == MF16) asm volatile(X8(MF16_4) ::: CLOB);
    if constexpr (CLS == MBF16) asm volatile(X8(MBF16_4) ::: CLOB);
    if constexpr (CLS == MF32) asm volatile(X8(MF32_4) ::: CLOB);
}

template <int CLS, int PRIO>
__device__ __forceinline__ unsigned run_stream(unsigned long long t0, unsigned long long budget)
{
    unsigned n = 0;
    if constexpr (CLS == IDLE) return 0;
    __builtin_amdgcn_s_setprio(PRIO);
    do {
#pragma unroll
        for (int r = 0; r < REPS; ++r) group<CLS>();
        n += REPS;
    } while (__builtin_amdgcn_s_memtime() - t0 < budget);
    __builtin_amdgcn_s_setprio(0);
    return n;
}

template <int CA, int PA, int CB, int PB>
__global__ __launch_bounds__(512) void pair_kernel(float *out, unsigned long long *rec, float seed, unsigned long long budget)
{
    asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0\n" ::"v"(seed * 1e-3f) : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned n;
    if (__builtin_amdgcn_readfirstlane(wv) < 4) n = run_stream<CA, PA>(t0, budget);
    else n = run_stream<CB, PB>(t0, budget);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_add_f32 %0, v100, v177" : "=v"(s));
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *p = rec + ((size_t)blockIdx.x * 8 + wv) * 4;
        p[0] = n; p[1] = t1 - t0; p[2] = r1 - r0; p[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    }
}

static const char *NAME[] = {"v_fma_f32 (3 src)", "v_add_f32 (2 src)", "v_add_f32_dpp", "v_exp_f32", "idle", "mfma_16x16x16_f16", "mfma_32x32x16_bf16", "mfma_16x16x4_f32"};
static double SOLO[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // same-stream price per instruction at two waves per SIMD, filled by the first runs

template <int CA, int PA, int CB, int PB>
int run()
{
    const int blocks = 256;
    const unsigned long long budget = 3000000ull;
    float *d; unsigned long long *c;
    CK(hipMalloc(&d, (size_t)blocks * 512 * 4)); CK(hipMalloc(&c, (size_t)blocks * 8 * 32));
    auto kern = pair_kernel<CA, PA, CB, PB>;
    const size_t lds = 100 * 1024;                                                         // (more than half a CU's LDS: ONE workgroup per CU)
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d, c, 1.0f, budget / 8);    // warm-up
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d, c, 1.0f, budget);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)blocks * 8 * 4);
    CK(hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ca, cb, mhz;
    int paired = 0, pairs = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 4; ++w) {
            const unsigned long long *a = &h[((size_t)b * 8 + w) * 4], *q = &h[((size_t)b * 8 + w + 4) * 4];
            ++pairs;
            if (((a[3] >> 4) & 3) != ((q[3] >> 4) & 3)) continue;                          // not on one SIMD: skip the pair
            ++paired;
            // SIMD cycles per instruction each stream retired (both ran for the budget, give or take one trip of 4 groups)
            if (a[0]) ca.push_back((double)a[1] / ((double)a[0] * group_size(CA)));
            if (q[0]) cb.push_back((double)q[1] / ((double)q[0] * group_size(CB)));
            mhz.push_back((double)a[1] / (double)a[2] * 100.0);
        }
    auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double pa = med(ca), pb = med(cb), clk = med(mhz);
    // pa = SIMD cycles per instruction retired by stream A (1/pa = A's instructions per cycle), likewise pb
    printf("A: %-18s prio %d | B: %-18s prio %d | %d/%d pairs on one SIMD | A every %6.2f cyc, B every %6.2f cyc", NAME[CA], PA, NAME[CB], PB, paired, pairs, pa, pb);
    if (CA == CB && PA == PB) {
        SOLO[CA] = 1.0 / (1.0 / pa + 1.0 / pb);                                            // the SIMD's cycles per instruction, both waves' instructions counted
        printf(" | same-stream price %.2f cycles per instruction", SOLO[CA]);
    } else if (CB != IDLE && SOLO[CB] > 0 && pb > 0) {
        // the pair retires 1/pa + 1/pb instructions per cycle; with B's instructions charged at B's same-stream price, what is left
        // of each cycle pays for A's: price(A beside B) = (1 - SOLO[B] / pb) * pa
        printf(" | pair: %.2f cycles per instruction | price of A beside B: %.2f (A's same-stream price %.2f)",
               1.0 / (1.0 / pa + 1.0 / pb), (1.0 - SOLO[CB] / pb) * pa, SOLO[CA]);
    }
    printf(" | %.0f MHz\n", clk);
    CK(hipFree(d)); CK(hipFree(c));
    return 0;
}

int main()
{
    run<FMA3, 0, FMA3, 0>(); run<ADD2, 0, ADD2, 0>();
    // matrix instructions: two equal streams, one wave alone, and beside the other wave's vector instructions
    run<MF16, 0, MF16, 0>(); run<MF16, 0, IDLE, 0>(); run<MF16, 0, FMA3, 0>(); run<MF16, 1, FMA3, 0>(); run<MF16, 0, FMA3, 1>(); run<MF16, 0, ADD2, 0>();
    run<MBF16, 0, MBF16, 0>(); run<MBF16, 0, IDLE, 0>(); run<MBF16, 0, FMA3, 0>(); run<MBF16, 1, FMA3, 0>(); run<MBF16, 0, FMA3, 1>();
    run<MF32, 0, MF32, 0>(); run<MF32, 0, IDLE, 0>(); run<MF32, 0, FMA3, 0>(); run<MF32, 1, FMA3, 0>(); run<MF32, 0, FMA3, 1>();
    return 0;
}
