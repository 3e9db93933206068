// nlm_strip.hpp -- the NLM strip kernel (see nlm.hip for the algorithm) and its launcher, shared by the translation units that
// instantiate it: nlm.hip (the two tuned windows, the per-pixel fallback, the C-ABI), nlm_rt.hip (any search window, patches up to 9x9,
// strips of eight rows) and nlm_rt4.hip (patches of 10x10 .. 16x16, strips of four rows).  Three files so that the 134 instantiations
// compile in parallel.
#pragma once
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
    unsigned wg_first;     // this launch covers the workgroups [wg_first, wg_first + gridDim.x) of the tiles x frames grid (0: all of it)
    // accumulate mode (one dispatch of nonlocal.comp)
    const void *target;
    const void *neighbour;
    mid_weightinfo *W;
    int slo, shi;          // run-time search range of the RTS instantiations
    // fused temporal mode
    int n_frames, k, first, count;
    int out_u8;            // fused mode: outputs are RGBA8 frames (pack_rgba8) instead of float4
    int corunning;         // host side only: launches of the frame pipeline overlap each other (no HALF tail, nlm.hip)
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

// The vertical sums of HALF an 8-row strip -- output rows 0..3 (lower == false) or 4..7 (lower == true) from the 4 + PW - 1 rows they
// need -- with the additions of vertical_box<PW, 8> for those rows, in its order: the block decomposition is evaluated in the 8-row
// strip's frame with the rows outside this half left out (none of them feeds the half's outputs; they fold away).  Two waves that take
// one half each therefore produce the bits one wave produces for the whole strip.
template <int PW, bool LOWER>
__device__ __forceinline__ void vertical_box_half(const float (&D)[4 + PW - 1], float (&V)[4])
{
    constexpr int N8 = 8 + PW - 1;
    float D8[N8], V8[8];
    if constexpr (LOWER) {
#pragma unroll
        for (int m = 0; m < N8; ++m) D8[m] = (m >= 4) ? D[m - 4] : 0.f;
        vertical_box<PW, 8>(D8, V8);
#pragma unroll
        for (int k = 0; k < 4; ++k) V[k] = V8[k + 4];
    } else {
#pragma unroll
        for (int m = 0; m < N8; ++m) D8[m] = (m < 4 + PW - 1) ? D[m] : 0.f;
        vertical_box<PW, 8>(D8, V8);
#pragma unroll
        for (int k = 0; k < 4; ++k) V[k] = V8[k];
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

template <int SLO, int SHI, int PLO, int PHI, int R, int NW, int FMT, bool FUSED, bool MULTI, int U = 1, bool SYM = false, int SYP = 0, int PF = 0, bool HALF = false>
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
    // HALF: the launch shape for the last, partly filled round of a small launch.  Eight waves per workgroup on the SAME 32-row
    // tile, each taking half of an 8-row strip (R = 4; even waves the upper, odd waves the lower four rows) with the strip's own
    // vertical sums (vertical_box_half): identical output bits, 0.6 of a strip's instructions per wave, and two waves per SIMD on
    // a CU that holds this workgroup alone -- where a 4-wave workgroup alone leaves every wave a SIMD to itself at half issue rate.
    static_assert(!HALF || (R == 4 && NW == 8 && !SYM && SYP == 0 && PF == 0 && MID_NLM_WALK > 0), "HALF: eight waves of four rows on the walk loop");

    extern __shared__ float4 lds[];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned tiles = (unsigned)(a.tiles_x * a.tiles_y);
    const unsigned bid = blockIdx.x + a.wg_first;
    const int fz = (int)(bid / tiles);
    const unsigned trem = xcd_remap_in_frame(bid - (unsigned)fz * tiles, tiles, (unsigned)fz);
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
            constexpr bool EARLY = NL >= 1;
            const bool lower_half = HALF && (__builtin_amdgcn_readfirstlane(wv) & 1) != 0;      // (scalar: a real branch, not both sides under masks)     // the row that leaves the window is no centre row: its slot can be refilled right after the distance phase
            // one offset of a run: window row r lives in register slot (j + r) % DR
            // (LT: std::bool_constant -- the lower half of a strip in the HALF shape; the two halves are two copies of the loop, chosen
            // per wave by a scalar branch around a whole run, so that each copy is straight-line code)
            auto step = [&](auto LT, int j, float4 (&n)[DR], const float4 *nextp, bool more) {
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
                if constexpr (HALF) vertical_box_half<PW, decltype(LT)::value>(D, V);
                else vertical_box<PW, R>(D, V);
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
            auto run = [&](auto LT, const bool FULL, const float4 *colp, int steps) __attribute__((always_inline)) {
                float4 n[DR];
#pragma unroll
                for (int m = 0; m < DR; ++m) n[m] = colp[m * LW];
#pragma unroll
                for (int j = 0; j < WALK; ++j) {
                    if (FULL || j < steps) step(LT, j, n, colp + (DR + j) * LW, FULL ? j + 1 < WALK : j + 1 < steps);
                }
                // rows still in the window that never were centre rows: keep their alpha formally live (see above)
                const float last_w = accw[R - 1];
#pragma unroll
                for (int m = 0; m < DR; ++m) asm volatile("" ::"v"(n[m].w), "v"(last_w));
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
                    if (HALF && lower_half) {
                        if (steps == WALK) { for (int sx = 0; sx < SW; ++sx) run(std::bool_constant<HALF>{}, true, rowp + sx, steps); }
                        else { for (int sx = 0; sx < SW; ++sx) run(std::bool_constant<HALF>{}, false, rowp + sx, steps); }
                    } else {
                        if (steps == WALK) { for (int sx = 0; sx < SW; ++sx) run(std::false_type{}, true, rowp + sx, steps); }
                        else { for (int sx = 0; sx < SW; ++sx) run(std::false_type{}, false, rowp + sx, steps); }
                    }
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

template <int SLO, int SHI, int PLO, int PHI, int R, int NW, int FMT, bool FUSED, bool MULTI, int U = 1, bool SYM = false, int SYP = 0, int PF = 0, bool HALF = false>
static int launch_strip(mid_ctx *ctx, NlmArgs &a, hipStream_t s, unsigned wg_first = 0, unsigned wg_count = ~0u)
{
    constexpr bool RTS = (SLO == 0 && SHI == 0);
    constexpr int PW = PHI - PLO;
    constexpr int VW = 64 - (PW - 1), TILE_H = NW * R;
    const int SW = RTS ? a.shi - a.slo : SHI - SLO;
    const int LW = 64 + SW - 1, LH = TILE_H + PW - 1 + (SYP > 0 ? SYP : SW) - 1;
    const size_t lds_bytes = (size_t)LW * LH * sizeof(float4);
    auto kern = nlm_strip_kernel<SLO, SHI, PLO, PHI, R, NW, FMT, FUSED, MULTI, U, SYM, SYP, PF, HALF>;
    if ((int)lds_bytes > ctx->lds_max)
        return set_error(MID_ERR_UNSUPPORTED, "nlm tile needs %zu B of LDS, device offers %d", lds_bytes, ctx->lds_max);
    // run-time-range instantiations are launched with different tile sizes: raise their limit to the device maximum once
    if (int rc = ensure_lds(ctx, (const void *)kern, RTS ? (size_t)ctx->lds_max : lds_bytes)) return rc;
    a.tiles_x = (int)cdiv(a.w, VW);
    a.tiles_y = (int)cdiv(a.h, TILE_H);
    const unsigned nwg = (unsigned)a.tiles_x * a.tiles_y * (FUSED ? a.count : 1);
    // (a launch may cover a sub-range of the tiles x frames grid: the tail of a small launch goes to the HALF shape)
    if (wg_first >= nwg) return MID_OK;
    const unsigned n = wg_count < nwg - wg_first ? wg_count : nwg - wg_first;
    if (n == 0) return MID_OK;
    a.wg_first = wg_first;
    hipLaunchKernelGGL(kern, dim3(n), dim3(NW * 64), lds_bytes, s, a);
    a.wg_first = 0;
    MID_HIP(hipGetLastError());
    return MID_OK;
}

// Workgroups of a tuned launch (4 waves x 8 rows per tile): what the tail rule of dispatch_ranges needs before launching.
inline unsigned nlm_tile_workgroups(int w, int h, int patch_w, int frames)
{
    return cdiv((unsigned)w, (unsigned)(64 - (patch_w - 1))) * cdiv((unsigned)h, 32u) * (unsigned)frames;
}

// Run-time search windows: defined in nlm_rt.hip / nlm_rt4.hip.  *handled = false: no strip instantiation for this patch (or the tile
// does not fit the LDS) -- the caller falls back to the per-pixel kernel.
int nlm_dispatch_rt8(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, int fmt, bool fused, bool *handled);
int nlm_dispatch_rt4(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, int fmt, bool fused, bool *handled);

}  // namespace mid
