// microbench13.hip -- which element of the NLM offset loop keeps its DPP phase from overlapping with the other wave's plain
// instructions the way microbench12's does (48 DPP @prio 1 + 144 FMA @prio 0: 455 cycles against 551 for the parts)?
// Built up step by step towards the loop's shape: + 8 v_exp_f32 in the raised block; + 14 ds_read_b128 and their wait in front
// of the plain block; the plain block split 84 / 18 / 40 around the raised block as in the kernel (distance | vertical sums |
// DPP, exp | accumulate) with priority codes 00110 and 01111.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define F(a) "v_fma_f32 v" #a ", v80, v81, v" #a "\n"
#define D(a) "v_add_f32_dpp v" #a ", v" #a ", v80 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define E(a) "v_exp_f32 v" #a ", v81\n"
#define F6(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5)
#define F12(b) F(1##b##0) F(1##b##1) F(1##b##2) F(1##b##3) F(1##b##4) F(1##b##5) F(1##b##6) F(1##b##7) F(1##b##8) F(1##b##9) F(1##b##0) F(1##b##1)
#define F84 F12(0) F12(1) F12(2) F12(3) F12(4) F12(5) F12(6)
#define F18 F12(7) F6(0)
#define F40 F12(1) F12(2) F12(3) F(140) F(141) F(142) F(143)
#define F142 F84 F18 F40
#define D4(b) D(2##b##0) D(2##b##1) D(2##b##2) D(2##b##3)
#define D48 D4(0) D4(1) D4(2) D4(3) D4(4) D4(5) D4(0) D4(1) D4(2) D4(3) D4(4) D4(5)
#define E8 E(90) E(91) E(92) E(93) E(94) E(95) E(96) E(97)
// 14 LDS reads of 16 B per lane (conflict-free: consecutive lanes, consecutive float4), then the wait the kernel has
#define L(r, off) "ds_read_b128 v[" #r ":" #r "+3], v82 offset:" #off "\n"
#define L14 "ds_read_b128 v[20:23], v82\n ds_read_b128 v[24:27], v82 offset:1344\n ds_read_b128 v[28:31], v82 offset:2688\n ds_read_b128 v[32:35], v82 offset:4032\n" \
            "ds_read_b128 v[36:39], v82 offset:5376\n ds_read_b128 v[40:43], v82 offset:6720\n ds_read_b128 v[44:47], v82 offset:8064\n ds_read_b128 v[48:51], v82 offset:9408\n" \
            "ds_read_b128 v[52:55], v82 offset:10752\n ds_read_b128 v[56:59], v82 offset:12096\n ds_read_b128 v[60:63], v82 offset:13440\n ds_read_b128 v[64:67], v82 offset:14784\n" \
            "ds_read_b128 v[68:71], v82 offset:16128\n ds_read_b128 v[72:75], v82 offset:17472\n s_waitcnt lgkmcnt(0)\n"
#define P(n) "s_setprio " #n "\n"
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
  "v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75", \
  "v90","v91","v92","v93","v94","v95","v96","v97", \
  "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
  "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
  "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
  "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179", \
  "v200","v201","v202","v203","v210","v211","v212","v213","v220","v221","v222","v223","v230","v231","v232","v233","v240","v241","v242","v243","v250","v251","v252","v253"
#define PROBE(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *stamps, float seed)        \
    {                                                                                                      \
        extern __shared__ float4 lds[];                                                                    \
        for (int i = threadIdx.x; i < 1344 * 16 / 16; i += 256) lds[i] = make_float4(seed, 0.f, 0.f, 0.f); \
        __syncthreads();                                                                                   \
        asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_lshlrev_b32 v82, 4, %1\n" :: "v"(seed * 1e-3f), "v"(threadIdx.x & 63) : "v80", "v81", "v82"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) asm volatile(BODY ::: CLOB);                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s;                                                                                           \
        asm volatile("v_add_f32 %0, v100, v200\n v_add_f32 %0, %0, v177\n v_add_f32 %0, %0, v20" : "=v"(s)); \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long *p = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;            \
            p[0] = t1 - t0; p[1] = r1 - r0;                                                                \
        }                                                                                                  \
    }
PROBE(k_parts_f, F142)
PROBE(k_parts_d, D48)
PROBE(k_parts_e, E8)
PROBE(k_parts_l, L14)
PROBE(k_a0, D48 E8 F142)                                        // no priorities
PROBE(k_a1, P(1) D48 E8 P(0) F142)                              // raised block = DPP + exp, plain block contiguous
PROBE(k_b1, P(1) D48 E8 P(0) L14 F142)                          // + tile reads and their wait in front of the plain block
PROBE(k_c110, L14 F84 F18 P(1) D48 E8 P(0) F40)                 // the kernel's order, code 00110
PROBE(k_c1111, L14 F84 P(1) F18 D48 E8 F40 P(0))                // the kernel's order, code 01111
PROBE(k_c0, L14 F84 F18 D48 E8 F40)                             // the kernel's order, no priorities
#define L14NW "ds_read_b128 v[20:23], v82\n ds_read_b128 v[24:27], v82 offset:1344\n ds_read_b128 v[28:31], v82 offset:2688\n ds_read_b128 v[32:35], v82 offset:4032\n" \
            "ds_read_b128 v[36:39], v82 offset:5376\n ds_read_b128 v[40:43], v82 offset:6720\n ds_read_b128 v[44:47], v82 offset:8064\n ds_read_b128 v[48:51], v82 offset:9408\n" \
            "ds_read_b128 v[52:55], v82 offset:10752\n ds_read_b128 v[56:59], v82 offset:12096\n ds_read_b128 v[60:63], v82 offset:13440\n ds_read_b128 v[64:67], v82 offset:14784\n" \
            "ds_read_b128 v[68:71], v82 offset:16128\n ds_read_b128 v[72:75], v82 offset:17472\n"
#define WAIT "s_waitcnt lgkmcnt(0)\n"
PROBE(k_pf1, WAIT F84 L14NW P(1) F18 D48 E8 F40 P(0))           // reads of the next offset issued after the distance phase, waited for at the loop top
PROBE(k_pf2, WAIT F84 P(1) L14NW F18 D48 E8 F40 P(0))           // ... issued inside the raised block
PROBE(k_pf3, WAIT F84 P(1) F18 D48 L14NW E8 F40 P(0))           // ... after the DPP adds
PROBE(k_w1, L14 P(1) F84 F18 D48 E8 F40 P(0))                   // only the reads and their wait at low priority
PROBE(k_w2, P(1) L14NW P(0) WAIT P(1) F84 F18 D48 E8 F40 P(0))  // reads issued at raised priority, only the wait low
#define L6NW "ds_read_b128 v[20:23], v82\n ds_read_b128 v[24:27], v82 offset:1344\n ds_read_b128 v[28:31], v82 offset:2688\n" \
             "ds_read_b128 v[64:67], v82 offset:14784\n ds_read_b128 v[68:71], v82 offset:16128\n ds_read_b128 v[72:75], v82 offset:17472\n"
#define L8NW "ds_read_b128 v[32:35], v82 offset:4032\n ds_read_b128 v[36:39], v82 offset:5376\n ds_read_b128 v[40:43], v82 offset:6720\n ds_read_b128 v[44:47], v82 offset:8064\n" \
             "ds_read_b128 v[48:51], v82 offset:9408\n ds_read_b128 v[52:55], v82 offset:10752\n ds_read_b128 v[56:59], v82 offset:12096\n ds_read_b128 v[60:63], v82 offset:13440\n"
PROBE(k_pp, WAIT F84 L6NW P(1) F18 D48 E8 F40 P(0) L8NW)        // partial prefetch: the 6 halo rows after the distance phase, the 8 centre rows after the accumulate (no extra registers)
PROBE(k_pp2, WAIT F84 P(1) L6NW F18 D48 E8 F40 L8NW P(0))
#define L1W "ds_read_b128 v[20:23], v82\n s_waitcnt lgkmcnt(0)\n"
#define L2W "ds_read_b128 v[20:23], v82\n ds_read_b128 v[24:27], v82 offset:1344\n s_waitcnt lgkmcnt(0)\n"
#define L5W "ds_read_b128 v[20:23], v82\n ds_read_b128 v[24:27], v82 offset:1344\n ds_read_b128 v[28:31], v82 offset:2688\n ds_read_b128 v[32:35], v82 offset:4032\n ds_read_b128 v[36:39], v82 offset:5376\n s_waitcnt lgkmcnt(0)\n"
#define L3W "ds_read_b128 v[20:23], v82\n ds_read_b128 v[24:27], v82 offset:1344\n ds_read_b128 v[28:31], v82 offset:2688\n s_waitcnt lgkmcnt(0)\n"
PROBE(k_l1, L1W F84 P(1) F18 D48 E8 F40 P(0))                   // kernel order 01111 with ONE tile read per offset (search rows walked innermost: 13 of the 14 rows stay in registers)
PROBE(k_l2, L2W F84 P(1) F18 D48 E8 F40 P(0))
PROBE(k_l3, L3W F84 P(1) F18 D48 E8 F40 P(0))
PROBE(k_l5, L5W F84 P(1) F18 D48 E8 F40 P(0))
PROBE(k_c1111b, L14 F84 P(1) F18 D48 E8 F40 P(0) "s_nop 0\n")

template <typename K>
int run(const char *name, K kern)
{
    for (int wps : {2}) {
        const int blocks = 256 * wps;
        float *d; unsigned long long *c;
        CK(hipMalloc(&d, (size_t)blocks * 256 * 4)); CK(hipMalloc(&c, (size_t)blocks * 4 * 16));
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 76 * 1024, 0, d, c, 1.0f);     // 76 KB of LDS: two workgroups per CU, like the kernel
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 76 * 1024, 0, d, c, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
        CK(hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (size_t i = 0; i < h.size(); i += 2) mhz.push_back((double)h[i] / (double)h[i + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double clk = mhz[mhz.size() / 2];
        printf("%-58s waves/SIMD=%d  %.1f cycles per group per SIMD (wall %.3f ms at %.0f MHz)\n", name, wps, ms * 1e-3 * clk * 1e6 / ((double)ITERS * wps), ms, clk);
        CK(hipFree(d)); CK(hipFree(c));
    }
    return 0;
}
int main()
{
    run("142 fma", k_parts_f); run("48 dpp", k_parts_d); run("8 exp", k_parts_e); run("14 ds_read_b128 + wait", k_parts_l);
    run("[48 dpp, 8 exp][142 fma], no priorities", k_a0); run("[48 dpp, 8 exp]@1 [142 fma]@0", k_a1);
    run("[48 dpp, 8 exp]@1 [14 reads + wait, 142 fma]@0", k_b1);
    run("prefetch: reads after the distance phase, 01111", k_pf1); run("prefetch: reads first thing in the raised block", k_pf2);
    run("prefetch: reads after the DPP adds", k_pf3); run("partial prefetch: halo rows early, centre rows late", k_pp); run("partial prefetch, both issued at raised priority", k_pp2);
    run("only reads + wait low, everything else raised", k_w1);
    run("reads issued raised, only the wait low", k_w2);
    run("kernel order 01111, 1 read per offset", k_l1); run("kernel order 01111, 2 reads per offset", k_l2); run("kernel order 01111, 3 reads per offset", k_l3); run("kernel order 01111, 5 reads per offset", k_l5);
    run("kernel order, no priorities", k_c0); run("kernel order, 00110", k_c110); run("kernel order, 01111", k_c1111);
    return 0;
}
