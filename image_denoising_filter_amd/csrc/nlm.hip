// nlm.hip -- non-local means for gfx950 (replaces shaders/nonlocal.comp:28-72).
//
// Algorithm.  For target pixel p and search offset s the reference evaluates
//     d(p,s) = sum_{o in patch} |T(p+o) - Nb(p+s+o)|^2_rgb,   wt = exp(-d/h^2)
// by brute force (SW^2 * PW^2 texel pairs per pixel).  Here d(p,s) is the PW x PW box sum of
// the per-offset difference image D_s(q) = |T(q) - Nb(q+s)|^2, so every D_s(q) is computed
// once and shared by the PW^2 pixels whose patch covers q:
//   * a wave owns 64 adjacent columns x R rows; lane l owns one column, keeps its target
//     column strip T(q) in VGPRs for the whole kernel and reads Nb(q+s) from an LDS tile
//     (float4 per texel: one conflict-free ds_read_b128 per texel);
//   * the offsets are walked search row INNERMOST (MID_NLM_WALK): two offsets one search row apart share 13 of their
//     14 tile rows, which stay in a compile-time-indexed ring of registers -- 1.6 tile reads per offset instead of 14;
//   * each offset runs as phases -- distances | vertical sums | DPP sums | exps | accumulate -- and the wave raises its
//     issue priority (s_setprio) from the vertical sums on: on gfx950 a mix of one wave's plain instructions with the
//     other wave's DPP adds / transcendentals costs far more than its parts unless the DPP/exp wave has priority
//     (tools/microbench10-13.hip);
//   * the vertical PW-tap sums are formed in registers by block prefix/suffix sums (18 adds per 8 outputs);
//   * the horizontal PW-tap sums move across lanes with whole-wave DPP shifts fused into
//     v_add_f32 (no LDS traffic, no shuffles): 64-(PW-1) lanes hold finished patch distances;
//   * v_exp_f32 with the -log2(e)/h^2 factor folded in, then 4 FMAs + 1 add per (pixel,offset).
// All sums are of non-negative terms (no running-sum cancellation), so results agree with
// the reference order to a few ulp of the patch distance; the per-pixel sum over the offsets is taken column-outer /
// row-inner, not in the shader's y-outer order (same terms, different last bits; one order for every launch shape).
//
// Out-of-image texels are vec4(0) for both images (LDS halo zero-filled).
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#ifndef MID_NLM_MIN_WAVES
#define MID_NLM_MIN_WAVES 0      /* waves per SIMD the register allocator must leave room for: 0 = per instantiation (nlm_min_waves below), 1 / 2 = A/B builds */
#endif
#ifndef MID_NLM_WALK
#define MID_NLM_WALK 21          /* search rows walked innermost in runs of this many (0 = search column innermost, the round-1/2 order) */
#endif
#ifndef MID_NLM_SINGLE_SYP
#define MID_NLM_SINGLE_SYP 0     /* search rows per tile fill of the single-frame launches (0 = the single-pass tile for every launch size); must equal MID_NLM_WALK when both are set */
#endif
#ifndef MID_NLM_SINGLE_SYP_REF
#define MID_NLM_SINGLE_SYP_REF 0 /* the same for the reference's shipped windows */
#endif
#ifndef MID_NLM_DIST_SPLIT
#define MID_NLM_DIST_SPLIT 0
#endif
#ifndef MID_NLM_PRIO_PHASES
#define MID_NLM_PRIO_PHASES 1111 /* issue priority of the five phases of an offset, one decimal digit each: distance, vertical sums, DPP sums,
                                   exp, accumulate, preceded by an optional sixth digit for the issue of the next offset's tile reads; A/B builds pass other codes */
#endif

namespace mid {

struct NlmArgs {
    int w, h;
    float kexp;            // -log2(e) / h^2 (per-pixel fallback kernel)
    float sk, inv_sk;      // sqrt(log2(e))/h and its reciprocal: the strip kernels carry the exponent scale in the colours
    int tiles_x, tiles_y;
    // accumulate mode (one dispatch of nonlocal.comp)
    const void *target;
    const void *neighbour;
    mid_weightinfo *W;
    int slo, shi;          // run-time search range of the RTS instantiations
    // fused temporal mode
    int n_frames, k, first, count;
    int out_u8;            // fused mode: outputs are RGBA8 frames (pack_rgba8) instead of float4
    FrameTable frames;
    OutTable outs;
};

// V[k] = D[k] + ... + D[k+PW-1] for k = 0..R-1, with the block decomposition of van Herk / Gil-Werman:
// cut D into blocks of PW values, form running sums from each block's end (S) and from each block's start
// (Pf); a window that starts inside block b is S[k] (rest of block b) + Pf[k+PW-1] (head of block b+1), a
// window that starts on a block boundary is that block's total.  PW=7, R=8: 18 additions for 8 outputs
// instead of 36 with shared pair/quad sums (48 direct).  Every partial sum only ever adds non-negative
// terms, so there is no cancellation; unused S/Pf entries are dead code after unrolling.
template <int PW, int R>
__device__ __forceinline__ void vertical_box(const float (&D)[R + PW - 1], float (&V)[R])
{
    constexpr int N = R + PW - 1;
    float S[N], Pf[N];
#pragma unroll
    for (int m = N - 1; m >= 0; --m) {
        const bool block_end = (m % PW == PW - 1) || (m == N - 1);
        S[m] = block_end ? D[m] : D[m] + S[m + 1];
    }
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const bool block_start = (m % PW == 0);
        Pf[m] = block_start ? D[m] : Pf[m - 1] + D[m];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (k % PW == 0) V[k] = (k == 0) ? S[0] : Pf[k + PW - 1];
        else V[k] = S[k] + Pf[k + PW - 1];
    }
}

// H[l] = sum_{i=PLO}^{PHI-1} V[l+i] across lanes; valid for lanes -PLO .. 63-(PHI-1).
template <int PLO, int PHI>
__device__ __forceinline__ float horizontal_box(float v)
{
    constexpr int NL = -PLO, NR = PHI - 1;
    float c = v;
#pragma unroll
    for (int i = 0; i < NR; ++i) c = v + wave_shl1(c);          // v[l .. l+NR]
    if constexpr (NL > 0) {
        float b = v;
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) b = v + wave_shr1(b);  // v[l-(NL-1) .. l]
        c = c + wave_shr1(b);                                   // + v[l-NL .. l-1]
    }
    return c;
}

// SYM (tuning builds only, `make TUNING=1`, MID_NLM_VARIANT=7): the pair-symmetry ABLATION of DESIGN.md 3.1 -- for a
// single frame d(p,s) = d(p+s,-s), so only the "positive" half of the offsets is evaluated and each weight is applied
// twice: to p (as always) and, as w*T(p), to the partner pixel p+s through a mirror accumulator that moves one lane
// per search column (DPP shift fused into the add).  This build measures the INSTRUCTION-MIX cost only: the mirror
// sums are folded back into the wrong rows/lanes and contributions that would cross strip, wave-edge and tile borders
// are dropped, so its output is wrong by construction -- it is an upper bound on what pair sharing could reach here.
// SYP > 0: the search window is walked in passes of SYP search rows, the LDS tile holding only the rows one pass needs
// (TILE_H + PW-1 + SYP-1 instead of TILE_H + PW-1 + SW-1).  At 21x21/7x7 with SYP = 3 the tile is 84 x 40 texels = 52.5 KB,
// so THREE workgroups share a CU (3 waves per SIMD instead of 2; the kernel needs 166 VGPRs when asked to, no spill).  The
// offsets are visited in the same order (search row outer, search column inner), so the sums -- and the output bits --
// are those of the single-pass kernel.
// Waves per SIMD the register allocator must leave room for.  The LDS tile allows two, and the tuned kernels fit two without being
// asked (183-236 VGPRs).  The run-time-window TEMPORAL kernels carry the search range in registers and the per-frame totals on top: left
// alone, those with 4x4, 7x7, 8x8 and 9x9 patches took 256 VGPRs + 2..28 AGPRs = ONE wave per SIMD, at the single-wave issue rate
// (half the two-wave one, tools/microbench8.hip).  Asked for two they spill 24-100 bytes per lane of cold state instead: 13x13/9x9 k=2
// 1.41 -> 0.95 ms, 8x8/8x8 0.58 -> 0.37 ms per output frame.  The others are left unconstrained: the same request costs the 5x5-patch
// kernel 9 % (255 -> 247 VGPRs, a tighter schedule) and the rest 2 % (profiles/r03_nlm_runtime_windows.txt).
constexpr int nlm_min_waves(bool rts, bool multi, int pw)
{
    return MID_NLM_MIN_WAVES > 0 ? MID_NLM_MIN_WAVES : (rts && multi && (pw == 4 || pw >= 7)) ? 2 : 1;
}

template <int SLO, int SHI, int PLO, int PHI, int R, int NW, int FMT, bool FUSED, bool MULTI, int U = 1, bool SYM = false, int SYP = 0, int PF = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(SYP > 0 ? 3 : nlm_min_waves(SLO == 0 && SHI == 0, MULTI, PHI - PLO), SYP > 0 ? 3 : 2)))
void nlm_strip_kernel(const NlmArgs a)
{
    // SLO == SHI == 0 selects the run-time search range [a.slo, a.shi) (any window, same patch):
    // LDS pitch and loop bounds then come from the arguments instead of being folded constants.
    constexpr bool RTS = (SLO == 0 && SHI == 0);
    constexpr int PW = PHI - PLO;
    constexpr int DR = R + PW - 1;
    constexpr int NL = -PLO, NR = PHI - 1;
    constexpr int VW = 64 - (PW - 1);
    constexpr int TILE_H = NW * R;
    const int slo = RTS ? a.slo : SLO;
    const int SW = RTS ? a.shi - a.slo : SHI - SLO;
    const int LW = 64 + SW - 1;
    const int SYPASS = SYP > 0 ? SYP : SW;                       // search rows per tile fill
    const int LH = TILE_H + PW - 1 + SYPASS - 1;
    static_assert(PLO <= 0 && PHI >= 1 && (RTS || SHI - SLO >= 1), "ranges must contain 0");
    static_assert(!(SYP > 0 && SYM), "the symmetry ablation is single-pass");
    static_assert(PF == 0 || (!SYM && SYP == 0), "the prefetching loop exists for the plain single-pass kernel");

    extern __shared__ float4 lds[];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned tiles = (unsigned)(a.tiles_x * a.tiles_y);
    const int fz = (int)(blockIdx.x / tiles);
    const unsigned trem = xcd_remap_in_frame(blockIdx.x - (unsigned)fz * tiles, tiles, (unsigned)fz);
    const int ty = (int)(trem / (unsigned)a.tiles_x), tx = (int)(trem - (unsigned)ty * a.tiles_x);

    const int w = a.w, h = a.h;
    const int X0 = tx * VW, Y0 = ty * TILE_H;
    const int gx = X0 + PLO + lane;          // column owned by this lane
    const int yb = Y0 + wv * R;              // first output row of this wave
    const bool wave_active = yb < h;

    const int t_out = a.first + fz;          // FUSED: output frame
    const void *target = FUSED ? a.frames.p[t_out] : a.target;
    int f_lo = FUSED ? t_out : 0, f_hi = f_lo;
    if (FUSED && MULTI) {
        f_lo = t_out - a.k < 0 ? 0 : t_out - a.k;
        f_hi = t_out + a.k > a.n_frames - 1 ? a.n_frames - 1 : t_out + a.k;
    }

    // Target column strip, kept in registers for every offset and every neighbour frame.
    float Tr[DR], Tg[DR], Tb[DR];
#ifdef MID_NLM_PKD   // tuning experiment: red/green differences as one v_pk_add_f32
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f Trg[DR];
#endif
#pragma unroll
    for (int m = 0; m < DR; ++m) {
        // Colours are pre-multiplied by sqrt(log2(e))/h, so the patch distance IS the exp2 argument and the
        // multiply per (pixel, offset) disappears; the accumulated colours are unscaled once per frame.
        const float4 t = fetch_texture<FMT>(target, w, h, gx, yb + PLO + m);
        Tr[m] = t.x * a.sk; Tg[m] = t.y * a.sk; Tb[m] = t.z * a.sk;
#ifdef MID_NLM_PKD
        Trg[m] = v2f{Tr[m], Tg[m]};
#endif
    }

    float4 tot[R];
    float totw[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { tot[k] = make_float4(0.f, 0.f, 0.f, 0.f); totw[k] = 0.f; }

    for (int f = f_lo; f <= f_hi; ++f) {
        const void *nb = FUSED ? a.frames.p[f] : a.neighbour;
        __syncthreads();   // previous frame's readers are done with the tile
        fill_tile<FMT, false>(lds, LW, LH, nb, w, h, X0 + PLO + slo, Y0 + PLO + slo, tid, NW * 64, a.sk);
        __syncthreads();
        if (SYP == 0 && !wave_active) continue;   // (multi-pass: every wave must reach the barriers of the later passes)

        float4 acc[R];
        float accw[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); accw[k] = 0.001f; }  // nonlocal.comp:32-33

        // One search offset: n[m] = Nb(q + s) for the lane's DR rows -> distances -> box sums -> weights.
        //
        // Issue priority by phase (round 3).  The two waves of a SIMD are arbitrated by priority, then age.  Left alone, one
        // wave's plain instructions interleave with the other wave's DPP adds and transcendentals, and such a mix costs far
        // more than its parts: tools/microbench12.hip -- 48 DPP adds + 144 FMAs per wave take 656 cycles per group per SIMD
        // against 548 for the two blocks alone, and 455 with s_setprio raised around the DPP block; microbench10/11 show the
        // same for v_exp_f32.  So a wave raises its priority when it enters its DPP phase and drops it after its exps: it runs
        // through its expensive instructions in one piece while the other wave waits its turn, and the cheap phases pair up.
        // kPrio = priority of {distance, vertical sums, DPP sums, exp, accumulate}; scheduling barriers keep each phase in one
        // piece where the priority changes.  Same instructions, same order of operations per value: identical output bits.
        constexpr int kPrio[6] = {(MID_NLM_PRIO_PHASES / 10000) % 10, (MID_NLM_PRIO_PHASES / 1000) % 10, (MID_NLM_PRIO_PHASES / 100) % 10,
                                  (MID_NLM_PRIO_PHASES / 10) % 10, MID_NLM_PRIO_PHASES % 10,
                                  (MID_NLM_PRIO_PHASES / 100000) % 10};      // [5]: while the next offset's tile reads are issued
        auto phase = [&](auto from, auto to) {      // compile-time phase indices
            constexpr int a = kPrio[decltype(from)::value], b = kPrio[decltype(to)::value];
            if constexpr (a != b) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(b);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>; using P2 = std::integral_constant<int, 2>;
        using P3 = std::integral_constant<int, 3>; using P4 = std::integral_constant<int, 4>; using PL = std::integral_constant<int, 5>;
        auto compute = [&](const float4 (&n)[DR]) {
            phase(PL{}, P0{});
            float D[DR];
#pragma unroll
            for (int m = 0; m < DR; ++m) {
#if MID_NLM_DIST_SPLIT > 0      /* A/B builds: the last rows of the distance phase already at the next phase's priority */
                if (m == MID_NLM_DIST_SPLIT) phase(P0{}, P1{});
#endif
#ifdef MID_NLM_PKD
                const v2f d2 = Trg[m] - v2f{n[m].x, n[m].y};
                const float dx = d2.x, dy = d2.y, dz = Tb[m] - n[m].z;
#else
                const float dx = Tr[m] - n[m].x, dy = Tg[m] - n[m].y, dz = Tb[m] - n[m].z;
#endif
                D[m] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            }
#if !(MID_NLM_DIST_SPLIT > 0)
            phase(P0{}, P1{});
#endif
            float V[R];
            vertical_box<PW, R>(D, V);
            phase(P1{}, P2{});
            float dd[R], ww[R];
#pragma unroll
            for (int k = 0; k < R; ++k) dd[k] = horizontal_box<PLO, PHI>(V[k]);
            phase(P2{}, P3{});
            // (the builtin, not common.hpp's exp2_hw: in THIS loop a wait state after each v_exp_f32 measured 2 % slower as a
            // block of eight and 6 % slower exp by exp)
#pragma unroll
            for (int k = 0; k < R; ++k) ww[k] = __builtin_amdgcn_exp2f(-dd[k]);    // exp(-d/h^2), nonlocal.comp:55 (d carries log2(e)/h^2)
            phase(P3{}, P4{});
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float wt = ww[k];
                const float4 c = n[k + NL];                             // centre texel Nb(p+s) of output row k
                acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);   // :56
                acc[k].z = fmaf(c.z, wt, acc[k].z); acc[k].w = fmaf(c.w, wt, acc[k].w);
                accw[k] += wt;                                         // :57
            }
            // The halo rows' alpha is never used; without this the compiler narrows their loads to
            // ds_read_b96 (8 LDS cycles) instead of ds_read_b128 (4).  One empty asm at the END of the
            // offset (tied to the last accumulator so it cannot be hoisted) keeps them formally live
            // without putting a wait in front of the distance phase.
#pragma unroll
            for (int m = 0; m < DR; ++m)
                if (m < NL || m >= NL + R) asm volatile("" ::"v"(n[m].w), "v"(accw[R - 1]));
            phase(P4{}, PL{});
        };
        auto load = [&](float4 (&n)[DR], const float4 *p) {
#pragma unroll
            for (int m = 0; m < DR; ++m) n[m] = p[m * LW];
        };

        if constexpr (PF > 0) {
            // Software-pipelined tile reads (round 3).  In the plain loop an offset is load -> wait -> compute: its 14
            // ds_read_b128 are issued and the wave waits for them at once, the other wave of the SIMD covering the gap
            // alone -- at the single-wave issue rate (4.4 cycles per instruction against 2.2 for two waves,
            // tools/microbench8.hip).  Here the reads of offset o+1 are issued right after the DISTANCE phase of offset o:
            // the 6 patch-halo rows go back into the registers that phase has just finished with, the 8 centre rows (still
            // needed by o's accumulate step) into a second set (+32 VGPRs), and the box sums, exp and accumulate of o
            // (about 2/3 of an offset) run while they are in flight.  Arithmetic and order of operations are unchanged:
            // identical output bits.
            static_assert(DR - R == NL + NR, "halo rows");
            constexpr int NH = DR - R;                                   // patch-halo rows: NL above, NR below
            float4 hl[NH], c0[R], c1[R];
            auto load_rows = [&](float4 (&h)[NH], float4 (&c)[R], const float4 *p) {
#pragma unroll
                for (int m = 0; m < NL; ++m) h[m] = p[m * LW];
#pragma unroll
                for (int k = 0; k < R; ++k) c[k] = p[(NL + k) * LW];
#pragma unroll
                for (int m = 0; m < NR; ++m) h[NL + m] = p[(NL + R + m) * LW];
            };
            auto dist = [&](const float4 (&h)[NH], const float4 (&c)[R], float (&D)[DR]) {
#pragma unroll
                for (int m = 0; m < DR; ++m) {
                    const float4 &n = m < NL ? h[m] : (m < NL + R ? c[m - NL] : h[m - R]);
                    const float dx = Tr[m] - n.x, dy = Tg[m] - n.y, dz = Tb[m] - n.z;
                    D[m] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                }
                // (halo alpha is never used: keep it formally live up to here so its read stays a ds_read_b128)
#pragma unroll
                for (int m = 0; m < NH; ++m) asm volatile("" ::"v"(h[m].w));
            };
            auto finish = [&](const float (&D)[DR], const float4 (&c)[R]) {
                float V[R];
                vertical_box<PW, R>(D, V);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const float d = horizontal_box<PLO, PHI>(V[k]);
                    const float wt = __builtin_amdgcn_exp2f(-d);
                    acc[k].x = fmaf(c[k].x, wt, acc[k].x); acc[k].y = fmaf(c[k].y, wt, acc[k].y);
                    acc[k].z = fmaf(c[k].z, wt, acc[k].z); acc[k].w = fmaf(c[k].w, wt, acc[k].w);
                    accw[k] += wt;
                }
            };
            const float4 *base = lds + (wv * R) * LW + lane;
            const int n_off = SW * SW;
            auto ptr_of = [&](int o) { const int sy = o / SW; return base + sy * LW + (o - sy * SW); };
            load_rows(hl, c0, ptr_of(0));
            // straight-line trips of two offsets (the centre-row sets alternate); every trip ends with offset o+2's rows in
            // flight, so the last trip's prefetch is clamped to the last offset (a harmless re-read) and an odd count ends
            // with one single-offset step
            const int n_pair = n_off / 2;
            for (int t = 0; t < n_pair; ++t) {
                const int o = 2 * t;
                {
                    float D[DR];
                    dist(hl, c0, D);
                    __builtin_amdgcn_sched_barrier(0);
                    load_rows(hl, c1, ptr_of(o + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    finish(D, c0);
                    __builtin_amdgcn_sched_barrier(0);      // (or the next offset's distance phase is hoisted up to the reads just issued)
                    __builtin_amdgcn_s_setprio(0);
                }
                {
                    float D[DR];
                    dist(hl, c1, D);
                    __builtin_amdgcn_sched_barrier(0);
                    load_rows(hl, c0, ptr_of(o + 2 < n_off ? o + 2 : n_off - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    finish(D, c1);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            if (n_off & 1) {
                float D[DR];
                dist(hl, c0, D);
                finish(D, c0);
            }
        } else if constexpr (SYM) {
            float Ta[R];                                   // alpha of the lane's own centre texels (T carries rgb only)
#pragma unroll
            for (int k = 0; k < R; ++k) Ta[k] = fetch_texture<FMT>(target, w, h, gx, yb + k).w;
            // mirror step: M moves one lane towards the partner column, then takes w * T(p)
            auto compute_sym = [&](const float4 (&n)[DR], float4 (&M)[R], float (&Mw)[R]) {
                float D[DR];
#pragma unroll
                for (int m = 0; m < DR; ++m) {
                    const float dx = Tr[m] - n[m].x, dy = Tg[m] - n[m].y, dz = Tb[m] - n[m].z;
                    D[m] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                }
                float V[R];
                vertical_box<PW, R>(D, V);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const float d = horizontal_box<PLO, PHI>(V[k]);
                    const float wt = __builtin_amdgcn_exp2f(-d);
                    const float4 c = n[k + NL];
                    acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);
                    acc[k].z = fmaf(c.z, wt, acc[k].z); acc[k].w = fmaf(c.w, wt, acc[k].w);
                    accw[k] += wt;
                    M[k].x = wave_shl1(M[k].x) + Tr[k + NL] * wt; M[k].y = wave_shl1(M[k].y) + Tg[k + NL] * wt;
                    M[k].z = wave_shl1(M[k].z) + Tb[k + NL] * wt; M[k].w = wave_shl1(M[k].w) + Ta[k] * wt;
                    Mw[k] = wave_shl1(Mw[k]) + wt;
                }
#pragma unroll
                for (int m = 0; m < DR; ++m)
                    if (m < NL || m >= NL + R) asm volatile("" ::"v"(n[m].w), "v"(accw[R - 1]));
            };
            {   // the zero offset has no partner
                float4 n[DR];
                load(n, lds + (wv * R - slo) * LW + lane - slo);
                compute(n);
            }
            auto fold = [&](float4 (&M)[R], float (&Mw)[R]) {   // ABLATION: folded back in place (the real thing needs an LDS flush per row)
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    acc[k].x += M[k].x; acc[k].y += M[k].y; acc[k].z += M[k].z; acc[k].w += M[k].w;
                    accw[k] += Mw[k];
                    M[k] = make_float4(0.f, 0.f, 0.f, 0.f); Mw[k] = 0.f;
                }
            };
            float4 M[R];
            float Mw[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { M[k] = make_float4(0.f, 0.f, 0.f, 0.f); Mw[k] = 0.f; }
            {   // search row 0: columns 1 .. shi-1 only
                const float4 *rowp = lds + (wv * R - slo) * LW + lane;
                for (int sx = 1 - slo; sx < SW; ++sx) {
                    float4 n[DR];
                    load(n, rowp + sx);
                    compute_sym(n, M, Mw);
                }
                fold(M, Mw);
            }
            for (int sy = 1 - slo; sy < SW; ++sy) {       // search rows 1 .. shi-1, every column
                const float4 *rowp = lds + (wv * R + sy) * LW + lane;
#pragma unroll U
                for (int sx = 0; sx < SW; ++sx) {
                    float4 n[DR];
                    load(n, rowp + sx);
                    compute_sym(n, M, Mw);
                }
                fold(M, Mw);
            }
        } else if constexpr (MID_NLM_WALK > 0 && (SYP == 0 || SYP == MID_NLM_WALK)) {   // (tuning variants with other pass sizes keep the old loop below)
            // Search rows walked INNERMOST in runs of MID_NLM_WALK (round 3).  Two offsets that differ by one search row read 13
            // of the same 14 tile rows (the lane's column, rows sy..sy+13 against sy+1..sy+14), so within a run only ONE new row
            // is read per offset -- into the register slot of the row that has just left the window, right after the distance
            // phase has used it for the last time -- instead of all 14: (14 + W - 1) / W tile reads per offset.  The register
            // window is a ring indexed at compile time (the run is fully unrolled).  The order of the offsets -- runs of W
            // search rows; inside a run search column outer, row inner -- is the same in the single-pass and in the multi-pass
            // tile (whose passes are the runs), so the two still give identical bits; it differs from rounds 1-2 (and from the
            // shader's y-outer loop, nonlocal.comp:36-38) in the order the 441 non-negative terms are added.
            constexpr int WALK = (!RTS && SHI - SLO < MID_NLM_WALK) ? SHI - SLO : MID_NLM_WALK;   // (a tuned window narrower than the run: one run per search column)
            static_assert(SYP == 0 || SYP == WALK, "the multi-pass tile's passes are the runs of the walk");
            constexpr bool EARLY = NL >= 1;     // the row that leaves the window is no centre row: its slot can be refilled right after the distance phase
            // one offset of a run: window row r lives in register slot (j + r) % DR
            auto step = [&](int j, float4 (&n)[DR], const float4 *nextp, bool more) {
                phase(PL{}, P0{});
                float D[DR];
#pragma unroll
                for (int m = 0; m < DR; ++m) {
                    const float4 &t = n[(j + m) % DR];
                    const float dx = Tr[m] - t.x, dy = Tg[m] - t.y, dz = Tb[m] - t.z;
                    D[m] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                }
                // the row that leaves the window: its alpha was never used unless it has been a centre row; keep it formally
                // live up to here so that every tile read stays a ds_read_b128, then reuse its slot for the row that enters
                if constexpr (EARLY) {
                    asm volatile("" ::"v"(n[j % DR].w));
                    if (more) n[j % DR] = nextp[0];
                }
                phase(P0{}, P1{});
                float V[R];
                vertical_box<PW, R>(D, V);
                phase(P1{}, P2{});
                float dd[R], ww[R];
#pragma unroll
                for (int k = 0; k < R; ++k) dd[k] = horizontal_box<PLO, PHI>(V[k]);
                phase(P2{}, P3{});
#pragma unroll
                for (int k = 0; k < R; ++k) ww[k] = __builtin_amdgcn_exp2f(-dd[k]);
                phase(P3{}, P4{});
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const float wt = ww[k];
                    const float4 c = n[(j + k + NL) % DR];            // centre texel of output row k = window row k + NL
                    acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);
                    acc[k].z = fmaf(c.z, wt, acc[k].z); acc[k].w = fmaf(c.w, wt, acc[k].w);
                    accw[k] += wt;
                }
                if constexpr (!EARLY) { if (more) n[j % DR] = nextp[0]; }   // (patches that start at row 0: the leaving row was output row 0's centre)
                phase(P4{}, PL{});
            };
            // `steps` <= WALK consecutive search rows at one search column; FULL: steps == WALK is known at compile time
            auto run = [&](const bool FULL, const float4 *colp, int steps) __attribute__((always_inline)) {
                float4 n[DR];
#pragma unroll
                for (int m = 0; m < DR; ++m) n[m] = colp[m * LW];
#pragma unroll
                for (int j = 0; j < WALK; ++j) {
                    if (FULL || j < steps) step(j, n, colp + (DR + j) * LW, FULL ? j + 1 < WALK : j + 1 < steps);
                }
                // rows still in the window that never were centre rows: keep their alpha formally live (see above)
#pragma unroll
                for (int m = 0; m < DR; ++m) asm volatile("" ::"v"(n[m].w), "v"(accw[R - 1]));
            };
            for (int sy0 = 0; sy0 < SW; sy0 += WALK) {
                if constexpr (SYP > 0) {
                    if (sy0 > 0) {
                        __syncthreads();
                        fill_tile<FMT, false>(lds, LW, LH, nb, w, h, X0 + PLO + slo, Y0 + PLO + slo + sy0, tid, NW * 64, a.sk);
                        __syncthreads();
                    }
                }
                if (wave_active) {
                    const int steps = sy0 + WALK < SW ? WALK : SW - sy0;
                    const float4 *rowp = lds + (wv * R + (SYP > 0 ? 0 : sy0)) * LW + lane;
                    if (steps == WALK) { for (int sx = 0; sx < SW; ++sx) run(true, rowp + sx, steps); }
                    else { for (int sx = 0; sx < SW; ++sx) run(false, rowp + sx, steps); }
                }
            }
        } else if constexpr (SYP > 0) {
            for (int sy0 = 0; sy0 < SW; sy0 += SYP) {          // one tile fill per SYP search rows; same offset order as the single pass
                if (sy0 > 0) {
                    __syncthreads();
                    fill_tile<FMT, false>(lds, LW, LH, nb, w, h, X0 + PLO + slo, Y0 + PLO + slo + sy0, tid, NW * 64, a.sk);
                    __syncthreads();
                }
                if (wave_active) {
                    const int sy1 = sy0 + SYP < SW ? sy0 + SYP : SW;
                    for (int sy = sy0; sy < sy1; ++sy) {
                        const float4 *rowp = lds + (wv * R + sy - sy0) * LW + lane;
#pragma unroll U
                        for (int sx = 0; sx < SW; ++sx) {
                            float4 n[DR];
                            load(n, rowp + sx);
                            compute(n);
                        }
                    }
                }
            }
        } else
        for (int sy = 0; sy < SW; ++sy) {
            const float4 *rowp = lds + (wv * R + sy) * LW + lane;
#pragma unroll U
            for (int sx = 0; sx < SW; ++sx) {
                float4 n[DR];
                load(n, rowp + sx);
                compute(n);
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {   // nlmData[p] += ..., nonlocal.comp:61-62
            acc[k].x *= a.inv_sk; acc[k].y *= a.inv_sk; acc[k].z *= a.inv_sk;     // back to unscaled colours (alpha never was scaled)
            if (MULTI) {
                tot[k].x += acc[k].x; tot[k].y += acc[k].y; tot[k].z += acc[k].z; tot[k].w += acc[k].w;
                totw[k] += accw[k];
            } else {                    // 0 + x == x: a single frame's sums are the totals
                tot[k] = acc[k];
                totw[k] = accw[k];
            }
        }
        if (!MULTI) break;
    }

    if (wave_active && lane >= NL && lane <= 63 - NR && gx < w) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int gy = yb + k;
            if (gy >= h) break;
            const size_t idx = (size_t)gy * w + gx;
            if (FUSED) {
                float4 o;
                if (totw[k] == 0.0f) o = make_float4(1.f, 0.f, 1.f, 1.f);       // normalize.comp:36-38
                else o = make_float4(tot[k].x / totw[k], tot[k].y / totw[k], tot[k].z / totw[k], tot[k].w / totw[k]);
                if (a.out_u8) ((uint32_t *)a.outs.p[fz])[idx] = pack_rgba8(o);
                else ((float4 *)a.outs.p[fz])[idx] = o;
            } else {
                float4 *wp = (float4 *)(a.W + idx);
                float4 wc = wp[0], nw = wp[1];
                wc.x += tot[k].x; wc.y += tot[k].y; wc.z += tot[k].z; wc.w += tot[k].w;
                nw.x += totw[k];
                wp[0] = wc;
                wp[1] = nw;
            }
        }
    }
}

// Any other search/patch ranges: one thread per pixel, straight from the shader text
// (nonlocal.comp:36-59) with global-memory fetches.  Correct for every legal parameter set;
// not a tuned path.
template <int FMT, bool FUSED>
__global__ __launch_bounds__(256) void nlm_generic_kernel(const NlmArgs a, int slo, int shi, int plo, int phi)
{
    const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    const int fz = blockIdx.z, t_out = a.first + fz;
    const void *target = FUSED ? a.frames.p[t_out] : a.target;
    int f_lo = 0, f_hi = 0;
    if (FUSED) {
        f_lo = t_out - a.k < 0 ? 0 : t_out - a.k;
        f_hi = t_out + a.k > a.n_frames - 1 ? a.n_frames - 1 : t_out + a.k;
    }
    const bool inside = px < a.w && py < a.h;
    if (!inside) return;
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    float totw = 0.f;
    for (int f = f_lo; inside && f <= f_hi; ++f) {
        const void *nb = FUSED ? a.frames.p[f] : a.neighbour;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float accw = 0.001f;
        for (int y = py + slo; y < py + shi; ++y)
            for (int x = px + slo; x < px + shi; ++x) {
                float d = 0.f;
                for (int j = plo; j < phi; ++j)
                    for (int i = plo; i < phi; ++i) {
                        const float4 t = fetch_texture<FMT>(target, a.w, a.h, px + i, py + j);
                        const float4 n = fetch_texture<FMT>(nb, a.w, a.h, x + i, y + j);
                        const float dx = t.x - n.x, dy = t.y - n.y, dz = t.z - n.z;
                        d += fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    }
                const float wt = __builtin_amdgcn_exp2f(d * a.kexp);
                const float4 c = fetch_texture<FMT>(nb, a.w, a.h, x, y);
                acc.x = fmaf(c.x, wt, acc.x); acc.y = fmaf(c.y, wt, acc.y);
                acc.z = fmaf(c.z, wt, acc.z); acc.w = fmaf(c.w, wt, acc.w);
                accw += wt;
            }
        tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w;
        totw += accw;
    }
    if (inside) {
        const size_t idx = (size_t)py * a.w + px;
        if (FUSED) {
            float4 o;
            if (totw == 0.0f) o = make_float4(1.f, 0.f, 1.f, 1.f);
            else o = make_float4(tot.x / totw, tot.y / totw, tot.z / totw, tot.w / totw);
            if (a.out_u8) ((uint32_t *)a.outs.p[fz])[idx] = pack_rgba8(o);
            else ((float4 *)a.outs.p[fz])[idx] = o;
        } else {
            float4 *wp = (float4 *)(a.W + idx);
            float4 wc = wp[0], nw = wp[1];
            wc.x += tot.x; wc.y += tot.y; wc.z += tot.z; wc.w += tot.w;
            nw.x += totw;
            wp[0] = wc; wp[1] = nw;
        }
    }
}

template <int SLO, int SHI, int PLO, int PHI, int R, int NW, int FMT, bool FUSED, bool MULTI, int U = 1, bool SYM = false, int SYP = 0, int PF = 0>
static int launch_strip(mid_ctx *ctx, NlmArgs &a, hipStream_t s)
{
    constexpr bool RTS = (SLO == 0 && SHI == 0);
    constexpr int PW = PHI - PLO;
    constexpr int VW = 64 - (PW - 1), TILE_H = NW * R;
    const int SW = RTS ? a.shi - a.slo : SHI - SLO;
    const int LW = 64 + SW - 1, LH = TILE_H + PW - 1 + (SYP > 0 ? SYP : SW) - 1;
    const size_t lds_bytes = (size_t)LW * LH * sizeof(float4);
    auto kern = nlm_strip_kernel<SLO, SHI, PLO, PHI, R, NW, FMT, FUSED, MULTI, U, SYM, SYP, PF>;
    if ((int)lds_bytes > ctx->lds_max)
        return set_error(MID_ERR_UNSUPPORTED, "nlm tile needs %zu B of LDS, device offers %d", lds_bytes, ctx->lds_max);
    // run-time-range instantiations are launched with different tile sizes: raise their limit to the device maximum once
    if (int rc = ensure_lds(ctx, (const void *)kern, RTS ? (size_t)ctx->lds_max : lds_bytes)) return rc;
    a.tiles_x = (int)cdiv(a.w, VW);
    a.tiles_y = (int)cdiv(a.h, TILE_H);
    const unsigned nwg = (unsigned)a.tiles_x * a.tiles_y * (FUSED ? a.count : 1);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds_bytes, s, a);
    MID_HIP(hipGetLastError());
    return MID_OK;
}

template <int FMT, bool FUSED>
static int dispatch_ranges(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s)
{
    // Tile shapes were chosen by A/B on MI355X (tools/ab_nlm.py): 4 waves x R rows per
    // workgroup (R=8: 76 KB of LDS), so two workgroups share a CU and one computes while the other
    // refills its tile; the search-column loop is unrolled 3x (21 = 7*3) / 2x (14 = 7*2);
    // with the van Herk vertical sums 3x measured 2 % faster than 7x (3435 vs 3362 Mpixel/s, 8 frames).
    // The tile shape is the same for every launch size on purpose: the block-sum decomposition of
    // vertical_box makes the rounding of a pixel depend on its row within the strip, so a fixed R keeps
    // the output bits independent of batch size, sharding and fused-vs-dispatch-sequence (tested).  (A
    // shorter strip, R=6, filled the CUs better for ONE 1080p frame -- 0.69 vs 0.74 ms -- but would have
    // made single-frame and batched results differ in the last bit.)
    const bool multi = FUSED && a.k > 0;
#ifdef MID_NLM_TUNING   // `make TUNING=1`: extra tile shapes selectable per process for tools/ab_nlm.py; not in the shipped library
    static const int variant = getenv("MID_NLM_VARIANT") ? atoi(getenv("MID_NLM_VARIANT")) : 0;
#endif
    if (p->search_lo == -10 && p->search_hi == 11 && p->patch_lo == -3 && p->patch_hi == 4) {   // 21x21 / 7x7 (benchmark)
#ifdef MID_NLM_TUNING
        if (multi && variant == 1) return launch_strip<-10, 11, -3, 4, 8, 8, FMT, FUSED, FUSED, 1>(ctx, a, s);
        if (!multi && variant == 1) return launch_strip<-10, 11, -3, 4, 8, 8, FMT, FUSED, false, 3>(ctx, a, s);
        if (!multi && variant == 2) return launch_strip<-10, 11, -3, 4, 7, 12, FMT, FUSED, false, 3>(ctx, a, s);
        if (!multi && variant == 5) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 7>(ctx, a, s);
        if (!multi && variant == 7) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 1, true>(ctx, a, s);   // pair-symmetry ablation
        if (!multi && variant == 8) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 3, true>(ctx, a, s);
        if (variant == 9) {   // three workgroups per CU: 7 passes of 3 search rows (52.5 KB tile)
            if (multi) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, FUSED, 3, false, 3>(ctx, a, s);
            return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 3, false, 3>(ctx, a, s);
        }
        if (variant == 10) {  // the same passes, 7 search columns unrolled
            if (multi) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, FUSED, 7, false, 3>(ctx, a, s);
            return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 7, false, 3>(ctx, a, s);
        }
        // six-wave workgroups (round 3): two workgroups per CU are already 3 waves/SIMD, with 5 (SYP = 5: 84 x 58 texels,
        // 76.1 KB) or 3 (SYP = 7: 84 x 60, 78.8 KB) tile fills per frame instead of the 7 of variants 9/10
        if (!multi && variant == 11) return launch_strip<-10, 11, -3, 4, 8, 6, FMT, FUSED, false, 3, false, 5>(ctx, a, s);
        if (!multi && variant == 12) return launch_strip<-10, 11, -3, 4, 8, 6, FMT, FUSED, false, 3, false, 7>(ctx, a, s);
        if (!multi && variant == 13) return launch_strip<-10, 11, -3, 4, 8, 6, FMT, FUSED, false, 7, false, 7>(ctx, a, s);
        if (variant == 14) {  // software-pipelined tile reads (PF)
            if (multi) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, FUSED, 1, false, 0, 1>(ctx, a, s);
            return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 1, false, 0, 1>(ctx, a, s);
        }
#endif
        if (multi) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, FUSED, 3>(ctx, a, s);
#if MID_NLM_SINGLE_SYP > 0
        // A launch over ONE frame (the frame pipeline's launches, mid_nlm_accum, latency-bound callers) is 1156 workgroups: on
        // the 512 slots of the 76 KB single-pass tile that is 2.26 rounds, the last one a quarter full.  The multi-pass tile
        // (52.5 KB, three workgroups per CU, 768 slots: 1.5 rounds) walks the same offsets in the same order -- identical
        // output bits (tested) -- and finishes a lone frame 10 % sooner; over many frames the single-pass tile is as fast or
        // faster, so the choice goes by launch size.
        if (a.count == 1) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 3, false, MID_NLM_SINGLE_SYP>(ctx, a, s);
#endif
        return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false, 3>(ctx, a, s);
    }
    if (p->search_lo == -7 && p->search_hi == 7 && p->patch_lo == -3 && p->patch_hi == 3) {     // nonlocal.comp:5-6 as shipped
        if (multi) return launch_strip<-7, 7, -3, 3, 8, 4, FMT, FUSED, FUSED, 2>(ctx, a, s);
#if MID_NLM_SINGLE_SYP_REF > 0
        if (a.count == 1) return launch_strip<-7, 7, -3, 3, 8, 4, FMT, FUSED, false, 2, false, MID_NLM_SINGLE_SYP_REF>(ctx, a, s);   // 77 x 43 texels = 51.7 KB: three workgroups per CU
#endif
        return launch_strip<-7, 7, -3, 3, 8, 4, FMT, FUSED, false, 2>(ctx, a, s);
    }
    // Any other search window with one of the common patches: the same strip kernel with the search
    // range as a run-time argument (LDS pitch no longer a folded constant: a few % slower).
    a.slo = p->search_lo; a.shi = p->search_hi;
    {
        const int sw = p->search_hi - p->search_lo;
        auto fits = [&](int pw_) { return (size_t)(64 + sw - 1) * (32 + pw_ - 1 + sw - 1) * sizeof(float4) <= (size_t)ctx->lds_max; };
        // Windows from 23x23 up: the 4-wave tile passes 80 KB, ONE workgroup fits a CU and its four waves have a SIMD each -- the
        // single-wave issue rate, half the two-wave one.  An 8-wave workgroup (64 rows, one tile of up to 160 KB) brings the second
        // wave per SIMD back: 25x25/7x7 1.22 -> 0.75 ms, 31x31/7x7 1.88 -> 1.16 ms per 1080p frame (profiles/r03_nlm_runtime_windows.txt).
        // Strips stay 8 rows at multiples of 8: identical output bits (tested against the 4-wave shape).  Only the symmetric
        // 3x3 / 5x5 / 7x7 patches: every instantiation costs build time.
        auto tile_bytes = [&](int nw, int pw_) { return (size_t)(64 + sw - 1) * (nw * 8 + pw_ - 1 + sw - 1) * sizeof(float4); };
        auto wants8 = [&](int pw_) { return 2 * tile_bytes(4, pw_) > (size_t)ctx->lds_max && tile_bytes(8, pw_) <= (size_t)ctx->lds_max; };
#define MID_NLM_RT8(PLO_, PHI_)                                                                             \
        if (p->patch_lo == (PLO_) && p->patch_hi == (PHI_) && wants8((PHI_) - (PLO_))) {                        \
            if (multi) return launch_strip<0, 0, PLO_, PHI_, 8, 8, FMT, FUSED, FUSED, 1>(ctx, a, s);            \
            return launch_strip<0, 0, PLO_, PHI_, 8, 8, FMT, FUSED, false, 1>(ctx, a, s);                       \
        }
        MID_NLM_RT8(-3, 4) MID_NLM_RT8(-2, 3) MID_NLM_RT8(-1, 2)
#undef MID_NLM_RT8
#define MID_NLM_RT(PLO_, PHI_)                                                                              \
        if (p->patch_lo == (PLO_) && p->patch_hi == (PHI_) && fits((PHI_) - (PLO_))) {                          \
            if (multi) return launch_strip<0, 0, PLO_, PHI_, 8, 4, FMT, FUSED, FUSED, 1>(ctx, a, s);            \
            return launch_strip<0, 0, PLO_, PHI_, 8, 4, FMT, FUSED, false, 1>(ctx, a, s);                       \
        }
        MID_NLM_RT(-3, 4) MID_NLM_RT(-3, 3) MID_NLM_RT(-2, 3) MID_NLM_RT(-1, 2) MID_NLM_RT(-4, 5)
        MID_NLM_RT(-2, 2) MID_NLM_RT(-4, 4)      // 4x4 and 8x8: the reference's half-open style ([-P,P), shaders/nonlocal.comp:42-44) at other sizes
        MID_NLM_RT(-1, 1) MID_NLM_RT(0, 1)       // 2x2 ([-1,1)) and the pixel-wise filter (1x1 patch: no box sums left, the same loop)
#undef MID_NLM_RT
    }
    dim3 grid(cdiv(a.w, 16), cdiv(a.h, 16), FUSED ? a.count : 1);
    hipLaunchKernelGGL((nlm_generic_kernel<FMT, FUSED>), grid, dim3(256), 0, s, a, p->search_lo, p->search_hi, p->patch_lo, p->patch_hi);
    MID_HIP(hipGetLastError());
    return MID_OK;
}

static int check_params(const mid_nlm_params *p)
{
    MID_REQUIRE(p != nullptr, "nlm: params is NULL");
    MID_REQUIRE(p->width > 0 && p->height > 0, "nlm: bad size %dx%d", p->width, p->height);
    MID_REQUIRE((long)p->width * p->height < (1l << 30), "nlm: image too large");
    MID_REQUIRE(p->filteringParameter > 0.f, "nlm: filteringParameter must be > 0");
    MID_REQUIRE(p->search_lo <= 0 && p->search_hi >= 1 && p->patch_lo <= 0 && p->patch_hi >= 1,
                "nlm: half-open ranges [lo,hi) must contain 0");
    MID_REQUIRE(p->search_hi - p->search_lo <= 64 && p->patch_hi - p->patch_lo <= 16, "nlm: window too large");
    MID_REQUIRE(p->format == MID_FMT_RGBA32F || p->format == MID_FMT_RGBA8, "nlm: unknown format %d", p->format);
    return MID_OK;
}

static float kexp_of(float hparam)
{
    return (float)(-1.4426950408889634 / ((double)hparam * (double)hparam));
}

static void set_scales(NlmArgs &a, float hparam)
{
    a.kexp = kexp_of(hparam);
    a.sk = (float)(sqrt(1.4426950408889634) / (double)hparam);
    a.inv_sk = (float)(1.0 / (double)a.sk);
}

}  // namespace mid

using namespace mid;

extern "C" int mid_nlm_accum(mid_ctx *ctx, const mid_nlm_params *p, const void *target,
                             const void *neighbour, mid_weightinfo *W, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = check_params(p)) return rc;
    MID_REQUIRE(target && neighbour && W, "nlm_accum: NULL image pointer");
    NlmArgs a{};
    a.w = p->width; a.h = p->height; set_scales(a, p->filteringParameter);
    a.target = target; a.neighbour = neighbour; a.W = W;
    a.n_frames = 1; a.k = 0; a.first = 0; a.count = 1;
    if (p->format == MID_FMT_RGBA8) return dispatch_ranges<MID_FMT_RGBA8, false>(ctx, p, a, b.s);
    return dispatch_ranges<MID_FMT_RGBA32F, false>(ctx, p, a, b.s);
}

extern "C" int mid_nlm_temporal(mid_ctx *ctx, const mid_nlm_params *p, const void *const *frames,
                                int n_frames, int k, int first, int count, mid_pixel *const *out,
                                void *stream)
{
    return mid::nlm_temporal_out(ctx, p, frames, n_frames, k, first, count, (void *const *)out, 0, stream);
}

int mid::nlm_temporal_out(mid_ctx *ctx, const mid_nlm_params *p, const void *const *frames, int n_frames, int k,
                          int first, int count, void *const *out, int out_u8, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = check_params(p)) return rc;
    MID_REQUIRE(frames && out, "nlm_temporal: NULL table");
    MID_REQUIRE(n_frames >= 1 && k >= 0 && count >= 1 && first >= 0 && first + count <= n_frames,
                "nlm_temporal: bad frame range (n=%d k=%d first=%d count=%d)", n_frames, k, first, count);
    // Output frames are processed in chunks so that chunk + halo fits the by-value frame table.
    const int max_chunk = kMaxFrames - 2 * k;
    MID_REQUIRE(max_chunk >= 1, "nlm_temporal: k=%d too large", k);
    for (int c0 = first; c0 < first + count; c0 += max_chunk) {
        const int cn = (first + count - c0) < max_chunk ? (first + count - c0) : max_chunk;
        const int lo = c0 - k < 0 ? 0 : c0 - k;
        const int hi = c0 + cn - 1 + k > n_frames - 1 ? n_frames - 1 : c0 + cn - 1 + k;
        NlmArgs a{};
        a.w = p->width; a.h = p->height; set_scales(a, p->filteringParameter);
        a.n_frames = hi - lo + 1; a.k = k; a.first = c0 - lo; a.count = cn; a.out_u8 = out_u8;
        for (int f = lo; f <= hi; ++f) {
            MID_REQUIRE(frames[f] != nullptr, "nlm_temporal: frame %d is NULL", f);
            a.frames.p[f - lo] = frames[f];
        }
        for (int t = 0; t < cn; ++t) {
            MID_REQUIRE(out[c0 - first + t] != nullptr, "nlm_temporal: out %d is NULL", c0 - first + t);
            a.outs.p[t] = out[c0 - first + t];
        }
        int rc = (p->format == MID_FMT_RGBA8) ? dispatch_ranges<MID_FMT_RGBA8, true>(ctx, p, a, b.s)
                                              : dispatch_ranges<MID_FMT_RGBA32F, true>(ctx, p, a, b.s);
        if (rc) return rc;
    }
    return MID_OK;
}
