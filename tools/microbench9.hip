// microbench9.hip -- do VGPR bank conflicts cost issue cycles on gfx950?  v_fma_f32 with explicit registers: the three
// sources in three different banks (register index mod 4), two in one bank, all three in one bank; and v_add_f32 likewise.
// Same harness as microbench8 (cycles from s_memtime, clock from s_memrealtime), 2 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 8192;
// 16 accumulators v[100..115]; operands in v[80..91] initialised from memory-independent values.
#define BODY16(F) F(100) F(101) F(102) F(103) F(104) F(105) F(106) F(107) F(108) F(109) F(110) F(111) F(112) F(113) F(114) F(115)
#define PROBE(NAME, INSTR)                                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n"      \
                     "v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0\n"      \
                     "v_mov_b32 v88, %0\n v_mov_b32 v89, %0\n v_mov_b32 v90, %0\n v_mov_b32 v91, %0\n" :: "v"(seed * 1e-3f) : "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91"); \
        asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n" \
                     "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n" \
                     ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                                               \
            asm volatile(INSTR INSTR INSTR INSTR INSTR INSTR INSTR INSTR                                   \
                         ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115"); \
        }                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v101\n v_add_f32 %0, %0, v102\n v_add_f32 %0, %0, v107\n v_add_f32 %0, %0, v115" : "=v"(s)); \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
// accumulators v100..v115: v100 is bank 0, v101 bank 1, ...  For accumulator A = 100+i (bank i%4):
//   "3 banks":   srcs in the two OTHER-bank registers chosen per accumulator bank
#define S2(x) #x
#define S(x) S2(x)
// helpers: for accumulator with bank b, pick operand registers: same bank -> v80+b (v80 b0, v81 b1, v82 b2, v83 b3), other banks -> v80+((b+1)%4), v84+((b+2)%4)
#define FMA_DIFF_0(a) "v_fma_f32 v" S(a) ", v81, v86, v" S(a) "\n"
#define FMA_DIFF_1(a) "v_fma_f32 v" S(a) ", v82, v87, v" S(a) "\n"
#define FMA_DIFF_2(a) "v_fma_f32 v" S(a) ", v83, v84, v" S(a) "\n"
#define FMA_DIFF_3(a) "v_fma_f32 v" S(a) ", v80, v85, v" S(a) "\n"
#define FMA_SAME2_0(a) "v_fma_f32 v" S(a) ", v80, v86, v" S(a) "\n"   /* src0 in the accumulator's bank */
#define FMA_SAME2_1(a) "v_fma_f32 v" S(a) ", v81, v87, v" S(a) "\n"
#define FMA_SAME2_2(a) "v_fma_f32 v" S(a) ", v82, v84, v" S(a) "\n"
#define FMA_SAME2_3(a) "v_fma_f32 v" S(a) ", v83, v85, v" S(a) "\n"
#define FMA_SAME3_0(a) "v_fma_f32 v" S(a) ", v80, v84, v" S(a) "\n"   /* all three in one bank */
#define FMA_SAME3_1(a) "v_fma_f32 v" S(a) ", v81, v85, v" S(a) "\n"
#define FMA_SAME3_2(a) "v_fma_f32 v" S(a) ", v82, v86, v" S(a) "\n"
#define FMA_SAME3_3(a) "v_fma_f32 v" S(a) ", v83, v87, v" S(a) "\n"
#define ADD_DIFF_0(a) "v_add_f32 v" S(a) ", v81, v" S(a) "\n"
#define ADD_DIFF_1(a) "v_add_f32 v" S(a) ", v82, v" S(a) "\n"
#define ADD_DIFF_2(a) "v_add_f32 v" S(a) ", v83, v" S(a) "\n"
#define ADD_DIFF_3(a) "v_add_f32 v" S(a) ", v80, v" S(a) "\n"
#define ADD_SAME_0(a) "v_add_f32 v" S(a) ", v80, v" S(a) "\n"
#define ADD_SAME_1(a) "v_add_f32 v" S(a) ", v81, v" S(a) "\n"
#define ADD_SAME_2(a) "v_add_f32 v" S(a) ", v82, v" S(a) "\n"
#define ADD_SAME_3(a) "v_add_f32 v" S(a) ", v83, v" S(a) "\n"
#define SEQ16(P) P##_0(100) P##_1(101) P##_2(102) P##_3(103) P##_0(104) P##_1(105) P##_2(106) P##_3(107) P##_0(108) P##_1(109) P##_2(110) P##_3(111) P##_0(112) P##_1(113) P##_2(114) P##_3(115)
PROBE(k_fma_diff, SEQ16(FMA_DIFF))
PROBE(k_fma_same2, SEQ16(FMA_SAME2))
PROBE(k_fma_same3, SEQ16(FMA_SAME3))
PROBE(k_add_diff, SEQ16(ADD_DIFF))
PROBE(k_add_same, SEQ16(ADD_SAME))

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
        const double n_instr = (double)ITERS * 8 * 16, clk = mhz[mhz.size() / 2];
        printf("%-34s waves/SIMD=%d  %.2f cyc per wave-instr per SIMD (wall %.3f ms at the measured %.0f MHz)\n", name, wps, ms * 1e-3 * clk * 1e6 / (n_instr * wps), ms, clk);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
int main()
{
    run("v_fma_f32, 3 sources in 3 banks", k_fma_diff); run("v_fma_f32, src0 in acc's bank", k_fma_same2); run("v_fma_f32, all 3 in one bank", k_fma_same3);
    run("v_add_f32, 2 sources in 2 banks", k_add_diff); run("v_add_f32, both in one bank", k_add_same);
    return 0;
}
