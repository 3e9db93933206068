// nlm_small.hip -- SMALL launches of the tuned NLM windows (a lone 1080p frame, two, three: at most kNlmSmallRounds rounds of workgroups).
//
// Same kernels, same bits as nlm.hip's (one template, csrc/nlm_strip.hpp; TAG = 1 only names the copy) -- compiled with another
// instruction-scheduling strategy.  The two-form offset loop (round 6) is fastest in long launches under LLVM's max-ILP strategy and fastest
// in launches of a few rounds -- where every workgroup is in the same phase at the same time -- under its iterative-ILP strategy: a lone
// 1080p frame 0.539 -> 0.521 ms, two 0.973 -> 0.960, three 1.376 -> 1.368; from four frames on max-ILP wins by 1 % (tools/frames_per_launch_ab.py,
// profiles/r06_ab_nlm_scheduling.txt, LABNOTES R6.9).  Scheduling moves no arithmetic (-ffp-contract=off): the launch size does not show in
// the output (tests: single == batched, HALF == whole strips).
//
// Also here, because only small launches use it: the HALF shape for the last round of a launch (see tail_split).
#include "nlm_strip.hpp"

namespace mid {

// The last round of a SMALL launch.  A tuned launch is tiles x frames workgroups of 4 waves, two per CU (76 KB tiles): `slots` at a
// time.  When the last round fills at most half of the CUs' slots -- one workgroup per CU or fewer -- every wave of it sits alone
// on its SIMD and issues at half rate, so the round takes as long as a full one: one 1080p frame is 1156 workgroups = 2.26 rounds and
// pays for 3.  Those workgroups are launched in the HALF shape instead (eight waves on the same tile, half a strip each, same bits:
// nlm_strip.hpp), which brings two waves per SIMD back and ends the round in 0.6 of the time.  Only for launches of a few rounds:
// in a long launch the last round is noise, and the headline launch stays ONE kernel.
static bool tail_split(unsigned slots, unsigned cu_count, unsigned nwg, unsigned &full, unsigned &rem)
{
    rem = nwg % slots;
    full = nwg - rem;
    return rem > 0 && rem <= cu_count;
}

template <int SLO, int SHI, int PLO, int PHI, int FMT, bool FUSED>
static int small_launch(mid_ctx *ctx, NlmArgs &a, hipStream_t s, unsigned slots, unsigned nwg)
{
    if (unsigned full, rem; tail_split(slots, (unsigned)ctx->cu_count, nwg, full, rem)) {
        if (int rc = launch_strip<SLO, SHI, PLO, PHI, 8, 4, FMT, FUSED, false, false, 1>(ctx, a, s, 0, full)) return rc;
        return launch_strip<SLO, SHI, PLO, PHI, 4, 8, FMT, FUSED, false, true, 1>(ctx, a, s, full, rem);
    }
    return launch_strip<SLO, SHI, PLO, PHI, 8, 4, FMT, FUSED, false, false, 1>(ctx, a, s);
}

template <int SLO, int SHI, int PLO, int PHI>
static int small_window(mid_ctx *ctx, NlmArgs &a, hipStream_t s, int fmt, bool fused, unsigned slots, unsigned nwg)
{
    if (fmt == MID_FMT_RGBA8) return fused ? small_launch<SLO, SHI, PLO, PHI, MID_FMT_RGBA8, true>(ctx, a, s, slots, nwg)
                                           : small_launch<SLO, SHI, PLO, PHI, MID_FMT_RGBA8, false>(ctx, a, s, slots, nwg);
    return fused ? small_launch<SLO, SHI, PLO, PHI, MID_FMT_RGBA32F, true>(ctx, a, s, slots, nwg)
                 : small_launch<SLO, SHI, PLO, PHI, MID_FMT_RGBA32F, false>(ctx, a, s, slots, nwg);
}

int nlm_dispatch_small(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s, int fmt, bool fused, bool *handled)
{
    *handled = false;
    if (a.corunning) return MID_OK;              // the frame pipeline keeps launches on two streams in flight: a long launch in effect
    const bool bench = p->search_lo == -10 && p->search_hi == 11 && p->patch_lo == -3 && p->patch_hi == 4;
    const bool ref = p->search_lo == -7 && p->search_hi == 7 && p->patch_lo == -3 && p->patch_hi == 3;
    if (!bench && !ref) return MID_OK;
    const unsigned slots = 2u * (unsigned)ctx->cu_count;
    const unsigned nwg = nlm_tile_workgroups(a.w, a.h, bench ? 7 : 6, fused ? a.count : 1);
    if (nwg > kNlmSmallRounds * slots) return MID_OK;
    *handled = true;
    return bench ? small_window<-10, 11, -3, 4>(ctx, a, s, fmt, fused, slots, nwg) : small_window<-7, 7, -3, 3>(ctx, a, s, fmt, fused, slots, nwg);
}

}  // namespace mid
