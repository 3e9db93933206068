// nlm_rt4.hip -- the NLM strip kernel with the search window as a run-time argument, patches 10x10 .. 16x16 (strips of four rows): shaders/nonlocal.comp:28-72 at other
// WINDOW / PATCH_WINDOW values than the shipped ones (:5-6), which nlm.hip's tuned instantiations serve.  Kernel and algorithm: nlm_strip.hpp, nlm.hip.
#include "nlm_strip.hpp"

namespace mid {

template <int FMT, bool FUSED>
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
            if (multi) return launch_strip<0, 0, PLO_, PHI_, 4, 4, FMT, FUSED, FUSED, 1>(ctx, a, s);            \
            return launch_strip<0, 0, PLO_, PHI_, 4, 4, FMT, FUSED, false, 1>(ctx, a, s);                       \
        }
        MID_NLM_RT4(-5, 5) MID_NLM_RT4(-5, 6) MID_NLM_RT4(-6, 6) MID_NLM_RT4(-6, 7) MID_NLM_RT4(-7, 7) MID_NLM_RT4(-7, 8) MID_NLM_RT4(-8, 8)
#undef MID_NLM_RT4
    }
    *handled = false;
    return MID_OK;
}

int nlm_dispatch_rt4(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, int fmt, bool fused, bool *handled)
{
    if (fmt == MID_FMT_RGBA8) return fused ? rt_ranges<MID_FMT_RGBA8, true>(ctx, p, a, s, handled) : rt_ranges<MID_FMT_RGBA8, false>(ctx, p, a, s, handled);
    return fused ? rt_ranges<MID_FMT_RGBA32F, true>(ctx, p, a, s, handled) : rt_ranges<MID_FMT_RGBA32F, false>(ctx, p, a, s, handled);
}

}  // namespace mid
