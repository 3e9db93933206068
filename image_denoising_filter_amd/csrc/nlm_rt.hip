// nlm_rt.hip -- the NLM strip kernel with the search window as a run-time argument, patches 1x1 .. 9x9 (strips of eight rows): shaders/nonlocal.comp:28-72 at other
// WINDOW / PATCH_WINDOW values than the shipped ones (:5-6), which nlm.hip's tuned instantiations serve.  Kernel and algorithm: nlm_strip.hpp, nlm.hip.
#include "nlm_strip.hpp"

namespace mid {

template <bool FUSED>
static int rt_ranges(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, bool *handled)
{
    const bool multi = FUSED && a.k > 0;
    *handled = true;
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
            if (multi) return launch_strip<0, 0, PLO_, PHI_, 8, 8, kFmtRuntime, FUSED, FUSED>(ctx, a, s);            \
            return launch_strip<0, 0, PLO_, PHI_, 8, 8, kFmtRuntime, FUSED, false>(ctx, a, s);                       \
        }
        MID_NLM_RT8(-3, 4) MID_NLM_RT8(-2, 3) MID_NLM_RT8(-1, 2)
#undef MID_NLM_RT8
#define MID_NLM_RT(PLO_, PHI_)                                                                              \
        if (p->patch_lo == (PLO_) && p->patch_hi == (PHI_) && fits((PHI_) - (PLO_))) {                          \
            if (multi) return launch_strip<0, 0, PLO_, PHI_, 8, 4, kFmtRuntime, FUSED, FUSED>(ctx, a, s);            \
            return launch_strip<0, 0, PLO_, PHI_, 8, 4, kFmtRuntime, FUSED, false>(ctx, a, s);                       \
        }
        MID_NLM_RT(-3, 4) MID_NLM_RT(-3, 3) MID_NLM_RT(-2, 3) MID_NLM_RT(-1, 2) MID_NLM_RT(-4, 5)
        MID_NLM_RT(-2, 2) MID_NLM_RT(-4, 4)      // 4x4 and 8x8: the reference's half-open style ([-P,P), shaders/nonlocal.comp:42-44) at other sizes
        MID_NLM_RT(-1, 1) MID_NLM_RT(0, 1)       // 2x2 ([-1,1)) and the pixel-wise filter (1x1 patch: no box sums left, the same loop)
#undef MID_NLM_RT
    }
    *handled = false;
    return MID_OK;
}

int nlm_dispatch_rt8(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, bool fused, bool *handled)
{
    // (the texel format is a kernel argument here, NlmArgs::fmt: a uniform branch around the tile fill and the target fetch instead
    // of a second set of instantiations -- half the build time and code size of these two files)
    return fused ? rt_ranges<true>(ctx, p, a, s, handled) : rt_ranges<false>(ctx, p, a, s, handled);
}

}  // namespace mid
