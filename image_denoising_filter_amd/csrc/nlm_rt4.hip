// nlm_rt4.hip -- the NLM strip kernel with the search window as a run-time argument, patches 10x10 .. 16x16 (strips of four rows): shaders/nonlocal.comp:28-72 at other
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
        // Patches of 10x10 .. 16x16 (the ABI's maximum): strips of FOUR rows (the lane's window is then 13-19 rows; 8 rows of a 7x7
        // patch take 14); each patch size has ONE strip height, so its bits are still independent of the launch shape.  Without these
        // the per-pixel fallback below does the naive S^2 P^2 work (5x5/11x11: 5.3 ms per 1080p frame).
        auto fits4 = [&](int pw_) { return (size_t)(64 + sw - 1) * (16 + pw_ - 1 + sw - 1) * sizeof(float4) <= (size_t)ctx->lds_max; };
#define MID_NLM_RT4(PLO_, PHI_)                                                                             \
        if (p->patch_lo == (PLO_) && p->patch_hi == (PHI_) && fits4((PHI_) - (PLO_))) {                         \
            if (multi) return launch_strip<0, 0, PLO_, PHI_, 4, 4, kFmtRuntime, FUSED, FUSED>(ctx, a, s);            \
            return launch_strip<0, 0, PLO_, PHI_, 4, 4, kFmtRuntime, FUSED, false>(ctx, a, s);                       \
        }
        MID_NLM_RT4(-5, 5) MID_NLM_RT4(-5, 6) MID_NLM_RT4(-6, 6) MID_NLM_RT4(-6, 7) MID_NLM_RT4(-7, 7) MID_NLM_RT4(-7, 8) MID_NLM_RT4(-8, 8)
#undef MID_NLM_RT4
    }
    *handled = false;
    return MID_OK;
}

int nlm_dispatch_rt4(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, bool fused, bool *handled)
{
    // (the texel format is a kernel argument here, NlmArgs::fmt: a uniform branch around the tile fill and the target fetch instead
    // of a second set of instantiations -- half the build time and code size of these two files)
    return fused ? rt_ranges<true>(ctx, p, a, s, handled) : rt_ranges<false>(ctx, p, a, s, handled);
}

}  // namespace mid
