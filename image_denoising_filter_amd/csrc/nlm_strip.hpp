// nlm_strip.hpp -- the NLM strip kernel (see nlm.hip for the algorithm) and its launcher, shared by the translation units that
// instantiate it: nlm.hip (the two tuned windows, the per-pixel fallback, the C-ABI), nlm_rt.hip (any search window, patches up to 9x9,
// strips of eight rows) and nlm_rt4.hip (patches of 10x10 .. 16x16, strips of four rows).  Three files so that the instantiations
// compile in parallel.  What was measured and rejected on the way here (pair symmetry, prefetching loops, multi-pass tiles, packed
// fp32, other priorities ...) is indexed in tools/experiments/README.md; none of it is in this file.
#pragma once
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace mid {

constexpr int kFmtRuntime = -1;  // FMT of the run-time-window instantiations: the texel format is the kernel argument NlmArgs::fmt
constexpr int kNlmWalk = 21;     // search rows walked innermost in runs of this many (measured against 3 and 7: profiles/r03_walk_ab.txt)

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
    int fmt;               // MID_FMT_* of the frames, read by the kFmtRuntime instantiations only
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

// Waves per SIMD the register allocator must leave room for.  The LDS tile allows two, and the tuned kernels fit two without being
// asked (183-236 VGPRs).  The run-time-window TEMPORAL kernels carry the search range in registers and the per-frame totals on top: left
// alone, those with 4x4, 7x7, 8x8 and 9x9 patches took 256 VGPRs + 2..28 AGPRs = ONE wave per SIMD, at the single-wave issue rate
// (half the two-wave one, tools/microbench8.hip).  Asked for two they spill 24-100 bytes per lane of cold state instead: 13x13/9x9 k=2
// 1.41 -> 0.95 ms, 8x8/8x8 0.58 -> 0.37 ms per output frame.  The others are left unconstrained: the same request costs the 5x5-patch
// kernel 9 % (255 -> 247 VGPRs, a tighter schedule) and the rest 2 % (profiles/r03_nlm_runtime_windows.txt).  Round 4: with the
// texel format as a run-time argument the 5x5-patch temporal kernel passes 256 as well (256 VGPRs + 1 AGPR: 15x15/5x5 k=2
// 0.83 -> 1.37 ms per output frame, profiles/r04_nlm_runtime_windows.txt), so it is asked too.
constexpr int nlm_min_waves(bool rts, bool multi, int pw)
{
    return (rts && multi && (pw == 4 || pw == 5 || pw >= 7)) ? 2 : 1;
}

// SLO == SHI == 0 selects the run-time search range [a.slo, a.shi) (any window, same patch): LDS pitch and loop bounds then come
// from the arguments instead of being folded constants.
// HALF: the launch shape for the last, partly filled round of a small launch.  Eight waves per workgroup on the SAME 32-row tile,
// each taking half of an 8-row strip (R = 4; even waves the upper, odd waves the lower four rows) with the strip's own vertical sums
// (vertical_box_half): identical output bits, 0.6 of a strip's instructions per wave, and two waves per SIMD on a CU that holds this
// workgroup alone -- where a 4-wave workgroup alone leaves every wave a SIMD to itself at half issue rate.
// TAG: no effect on the code -- it only names a second copy of an instantiation, so that the copies in nlm_small.hip (compiled with another
// scheduling strategy, see there) and in nlm.hip are different symbols.
template <int SLO, int SHI, int PLO, int PHI, int R, int NW, int FMT, bool FUSED, bool MULTI, bool HALF = false, int TAG = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(nlm_min_waves(SLO == 0 && SHI == 0, MULTI, PHI - PLO), 2)))
void nlm_strip_kernel(const NlmArgs a)
{
    constexpr bool RTS = (SLO == 0 && SHI == 0);
    constexpr int PW = PHI - PLO;
    constexpr int DR = R + PW - 1;
    constexpr int NL = -PLO, NR = PHI - 1;
    constexpr int VW = 64 - (PW - 1);
    constexpr int TILE_H = NW * R;
    const int slo = RTS ? a.slo : SLO;
    const int SW = RTS ? a.shi - a.slo : SHI - SLO;
    const int LW = 64 + SW - 1;
    const int LH = TILE_H + PW - 1 + SW - 1;
    static_assert(PLO <= 0 && PHI >= 1 && (RTS || SHI - SLO >= 1), "ranges must contain 0");
    static_assert(!HALF || (R == 4 && NW == 8), "HALF: eight waves of four rows");
    static_assert(FMT != kFmtRuntime || RTS, "only the run-time-window instantiations take the format as an argument");

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
    // Colours are pre-multiplied by sqrt(log2(e))/h, so the patch distance IS the exp2 argument and the
    // multiply per (pixel, offset) disappears; the accumulated colours are unscaled once per frame.
    float Tr[DR], Tg[DR], Tb[DR];
    if constexpr (FMT == kFmtRuntime) {          // (a uniform branch around two copies of the fetch loop / the fill loop)
        if (a.fmt == MID_FMT_RGBA8) {
#pragma unroll
            for (int m = 0; m < DR; ++m) {
                const float4 t = fetch_texture<MID_FMT_RGBA8>(target, w, h, gx, yb + PLO + m);
                Tr[m] = t.x * a.sk; Tg[m] = t.y * a.sk; Tb[m] = t.z * a.sk;
            }
        } else {
#pragma unroll
            for (int m = 0; m < DR; ++m) {
                const float4 t = fetch_texture<MID_FMT_RGBA32F>(target, w, h, gx, yb + PLO + m);
                Tr[m] = t.x * a.sk; Tg[m] = t.y * a.sk; Tb[m] = t.z * a.sk;
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < DR; ++m) {
            const float4 t = fetch_texture<FMT>(target, w, h, gx, yb + PLO + m);
            Tr[m] = t.x * a.sk; Tg[m] = t.y * a.sk; Tb[m] = t.z * a.sk;
        }
    }
    auto fill = [&](const void *nb, bool *opaque) {
        if constexpr (FMT == kFmtRuntime) {
            if (a.fmt == MID_FMT_RGBA8) fill_tile<MID_FMT_RGBA8, false>(lds, LW, LH, nb, w, h, X0 + PLO + slo, Y0 + PLO + slo, tid, NW * 64, a.sk, opaque);
            else fill_tile<MID_FMT_RGBA32F, false>(lds, LW, LH, nb, w, h, X0 + PLO + slo, Y0 + PLO + slo, tid, NW * 64, a.sk, opaque);
        } else {
            fill_tile<FMT, false>(lds, LW, LH, nb, w, h, X0 + PLO + slo, Y0 + PLO + slo, tid, NW * 64, a.sk, opaque);
        }
    };

    float4 tot[R];
    float totw[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { tot[k] = make_float4(0.f, 0.f, 0.f, 0.f); totw[k] = 0.f; }

    for (int f = f_lo; f <= f_hi; ++f) {
        const void *nb = FUSED ? a.frames.p[f] : a.neighbour;
        __syncthreads();   // previous frame's readers are done with the tile
        // Is every texel of the neighbour tile opaque (alpha == 1.0f -- the usual case away from the image border, where the tile
        // holds out-of-image texels, vec4(0))?  Then sum(wt * alpha) IS sum(wt) and the weight accumulator need not be carried
        // through the offsets: normWeight = 0.001 + weightColor.w at the end of the frame -- one add per output and offset less
        // (8 of 199 VALU) and eight registers free.  Round 6, the tuned windows only (kOpaqueForm): 31-frame launches +1.7 %, the
        // RGBA8 frame pipeline +2-3.5 %, temporal k = 2 +7-8 %.  A lone frame first LOST 3.6 % -- its main launch is exactly two rounds
        // of workgroups, all of them in the fill phase at once, where the agreement on `opaque` (a workgroup reduction) shows -- and
        // got it back from the compiler: launches of a few rounds run copies of these kernels scheduled by LLVM's iterative-ILP
        // strategy (nlm_small.hip), long ones the max-ILP copies (profiles/r06_ab_nlm_opaque_form.txt, r06_ab_nlm_scheduling.txt,
        // LABNOTES R6.8-R6.9).  The form is
        // chosen per workgroup and neighbour frame from the tile's CONTENT, and every launch shape of a window -- batched, single,
        // HALF tail, accumulate-only, temporal -- carries both forms, so a pixel's bits do not depend on the launch.  (In the opaque
        // form the 0.001 is added after the weights instead of before them: the last bit of normWeight, nothing else.)
        constexpr bool kOpaqueForm = !RTS;
        bool opaque = false;
        if constexpr (kOpaqueForm) {
            bool mine = true;
            fill(nb, &mine);
            opaque = __syncthreads_and(mine) != 0;
        } else {
            fill(nb, nullptr);
            __syncthreads();
        }
        if (!wave_active) continue;

        float4 acc[R];
        float accw[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); accw[k] = 0.001f; }  // nonlocal.comp:32-33

        // Issue priority by phase.  The two waves of a SIMD are arbitrated by priority, then age.  An offset runs as
        // distances | vertical sums | DPP sums | exps | accumulate; the wave raises its priority when it leaves the distance phase
        // and drops it before the next one, so it runs through its DPP adds and transcendentals in one piece while its partner,
        // if it is in its distance phase (tile reads, plain VALU), yields: a DPP add issued beside the OTHER wave's plain
        // instruction costs what a plain instruction costs, while beside DPP adds or behind transcendentals it costs twice that
        // (tools/microbench12.hip, tools/microbench17.hip; +15-21 % on this kernel, profiles/r03_ab_nlm_issue_priority.txt).
        // Scheduling barriers keep each phase in one piece where the priority changes.  Same instructions, same order of
        // operations per value: identical output bits.
        auto raise_priority = [] { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); __builtin_amdgcn_sched_barrier(0); };
        auto drop_priority = [] { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); };

        // Search rows walked INNERMOST in runs of kNlmWalk.  Two offsets that differ by one search row read DR - 1 of the same DR
        // tile rows (the lane's column, rows sy..sy+DR-1 against sy+1..sy+DR), so within a run only ONE new row is read per offset
        // -- into the register slot of the row that has just left the window, right after the distance phase has used it for the
        // last time -- instead of all DR: (DR + W - 1) / W tile reads per offset.  The register window is a ring indexed at
        // compile time (the run is fully unrolled).  The order of the offsets -- runs of W search rows; inside a run search column
        // outer, row inner -- differs from the shader's y-outer loop (nonlocal.comp:36-38) in the order the non-negative terms
        // of a pixel are added, i.e. in the last bits; every launch shape shares the one order.
        constexpr int WALK = (!RTS && SHI - SLO < kNlmWalk) ? SHI - SLO : kNlmWalk;   // (a tuned window narrower than the run: one run per search column)
        constexpr bool EARLY = NL >= 1;     // the row that leaves the window is no centre row: its slot can be refilled right after the distance phase
        const bool lower_half = HALF && (__builtin_amdgcn_readfirstlane(wv) & 1) != 0;      // (scalar: a real branch, not both sides under masks)
        // one offset of a run: window row r lives in register slot (j + r) % DR
        // (LT: std::bool_constant -- the lower half of a strip in the HALF shape; the two halves are two copies of the loop, chosen
        // per wave by a scalar branch around a whole run, so that each copy is straight-line code)
        auto step = [&](auto LT, auto A1, int j, float4 (&n)[DR], const float4 *nextp, bool more) {
            float D[DR];
#pragma unroll
            for (int m = 0; m < DR; ++m) {
                const float4 &t = n[(j + m) % DR];
                const float dx = Tr[m] - t.x, dy = Tg[m] - t.y, dz = Tb[m] - t.z;
                D[m] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            }
            // the row that leaves the window: its alpha was never used unless it has been a centre row; keep it formally
            // live up to here so that every tile read stays a ds_read_b128 (4 LDS cycles; a ds_read_b96 takes 8), then reuse
            // its slot for the row that enters
            if constexpr (EARLY) {
                asm volatile("" ::"v"(n[j % DR].w));
                if (more) n[j % DR] = nextp[0];
            }
            raise_priority();
            float V[R];
            if constexpr (HALF) vertical_box_half<PW, decltype(LT)::value>(D, V);
            else vertical_box<PW, R>(D, V);
            float dd[R], ww[R];
#pragma unroll
            for (int k = 0; k < R; ++k) dd[k] = horizontal_box<PLO, PHI>(V[k]);
            // (the builtin, not common.hpp's exp2_hw: in THIS loop a wait state after each v_exp_f32 measured 2 % slower as a
            // block of eight and 6 % slower exp by exp, LABNOTES.md)
#pragma unroll
            for (int k = 0; k < R; ++k) ww[k] = __builtin_amdgcn_exp2f(-dd[k]);    // exp(-d/h^2), nonlocal.comp:55 (d carries log2(e)/h^2)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float wt = ww[k];
                const float4 c = n[(j + k + NL) % DR];            // centre texel Nb(p+s) of output row k = window row k + NL
                acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);   // :56
                acc[k].z = fmaf(c.z, wt, acc[k].z); acc[k].w = fmaf(c.w, wt, acc[k].w);
                if constexpr (!decltype(A1)::value) accw[k] += wt;   // :57
            }
            if constexpr (!EARLY) { if (more) n[j % DR] = nextp[0]; }   // (patches that start at row 0: the leaving row was output row 0's centre)
            drop_priority();
        };
        // `steps` <= WALK consecutive search rows at one search column; FULL: steps == WALK is known at compile time
        auto run = [&](auto LT, auto A1, const bool FULL, const float4 *colp, int steps) __attribute__((always_inline)) {
            float4 n[DR];
#pragma unroll
            for (int m = 0; m < DR; ++m) n[m] = colp[m * LW];
#pragma unroll
            for (int j = 0; j < WALK; ++j) {
                if (FULL || j < steps) step(LT, A1, j, n, colp + (DR + j) * LW, FULL ? j + 1 < WALK : j + 1 < steps);
            }
            // rows still in the window that never were centre rows: keep their alpha formally live (see above)
            const float last_w = decltype(A1)::value ? acc[R - 1].w : accw[R - 1];
#pragma unroll
            for (int m = 0; m < DR; ++m) asm volatile("" ::"v"(n[m].w), "v"(last_w));
        };
        auto walk = [&](auto A1) {
            for (int sy0 = 0; sy0 < SW; sy0 += WALK) {
                const int steps = sy0 + WALK < SW ? WALK : SW - sy0;
                const float4 *rowp = lds + (wv * R + sy0) * LW + lane;
                if (HALF && lower_half) {
                    if (steps == WALK) { for (int sx = 0; sx < SW; ++sx) run(std::bool_constant<HALF>{}, A1, true, rowp + sx, steps); }
                    else { for (int sx = 0; sx < SW; ++sx) run(std::bool_constant<HALF>{}, A1, false, rowp + sx, steps); }
                } else {
                    if (steps == WALK) { for (int sx = 0; sx < SW; ++sx) run(std::false_type{}, A1, true, rowp + sx, steps); }
                    else { for (int sx = 0; sx < SW; ++sx) run(std::false_type{}, A1, false, rowp + sx, steps); }
                }
            }
        };
        if (kOpaqueForm && opaque) {
            walk(std::true_type{});
#pragma unroll
            for (int k = 0; k < R; ++k) accw[k] = 0.001f + acc[k].w;     // normWeight of this frame: nonlocal.comp:32 + sum(wt)
        } else {
            walk(std::false_type{});
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

template <int SLO, int SHI, int PLO, int PHI, int R, int NW, int FMT, bool FUSED, bool MULTI, bool HALF = false, int TAG = 0>
static int launch_strip(mid_ctx *ctx, NlmArgs &a, hipStream_t s, unsigned wg_first = 0, unsigned wg_count = ~0u)
{
    constexpr bool RTS = (SLO == 0 && SHI == 0);
    constexpr int PW = PHI - PLO;
    constexpr int VW = 64 - (PW - 1), TILE_H = NW * R;
    const int SW = RTS ? a.shi - a.slo : SHI - SLO;
    const int LW = 64 + SW - 1, LH = TILE_H + PW - 1 + SW - 1;
    const size_t lds_bytes = (size_t)LW * LH * sizeof(float4);
    auto kern = nlm_strip_kernel<SLO, SHI, PLO, PHI, R, NW, FMT, FUSED, MULTI, HALF, TAG>;
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
int nlm_dispatch_rt8(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, bool fused, bool *handled);
int nlm_dispatch_rt4(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, bool fused, bool *handled);
// Small launches of the tuned windows (at most kNlmSmallRounds rounds of workgroups, not co-running): defined in nlm_small.hip, which holds its
// own copies of the tuned kernels (TAG = 1) and the HALF shape for the last round.  *handled = false: not a tuned window / not a small launch.
constexpr unsigned kNlmSmallRounds = 7;
int nlm_dispatch_small(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, int fmt, bool fused, bool *handled);

}  // namespace mid
