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
//   * the offsets are walked search row INNERMOST (kNlmWalk, nlm_strip.hpp): two offsets one search row apart share 13 of their
//     14 tile rows, which stay in a compile-time-indexed ring of registers -- 1.6 tile reads per offset instead of 14;
//   * each offset runs as phases -- distances | vertical sums | DPP sums | exps | accumulate -- and the wave raises its
//     issue priority (s_setprio) from the vertical sums on: on gfx950 a mix of one wave's plain instructions with the
//     other wave's DPP adds / transcendentals costs far more than its parts unless the DPP/exp wave has priority
//     (tools/microbench10-13.hip, microbench17.hip);
//   * the vertical PW-tap sums are formed in registers by block prefix/suffix sums (18 adds per 8 outputs);
//   * the horizontal PW-tap sums move across lanes with whole-wave DPP shifts fused into
//     v_add_f32 (no LDS traffic, no shuffles): 64-(PW-1) lanes hold finished patch distances;
//   * v_exp_f32 with the -log2(e)/h^2 factor folded in, then 4 FMAs + 1 add per (pixel,offset).
// All sums are of non-negative terms (no running-sum cancellation), so results agree with
// the reference order to a few ulp of the patch distance; the per-pixel sum over the offsets is taken column-outer /
// row-inner, not in the shader's y-outer order (same terms, different last bits; one order for every launch shape).
//
// Out-of-image texels are vec4(0) for both images (LDS halo zero-filled).
#include "nlm_strip.hpp"
#include <unordered_set>

namespace mid {

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

template <int FMT, bool FUSED>
static int dispatch_ranges(mid_ctx *ctx, const mid_nlm_params *p, NlmArgs &a, hipStream_t s)
{
    // Tile shape (A/B on MI355X, LABNOTES.md): 4 waves x R = 8 rows per workgroup -- 76 KB of LDS, so two workgroups share a
    // CU and one computes while the other refills its tile.
    // The strip height is the same for every launch size on purpose: the block-sum decomposition of vertical_box makes the
    // rounding of a pixel depend on its row within the strip, so a fixed R keeps the output bits independent of batch size,
    // sharding and fused-vs-dispatch-sequence (tested).  The HALF shape (nlm_small.hip: tail_split) splits strips into rows 0-3 / 4-7
    // with the 8-row strip's own additions, so it may be used wherever it is faster.
    const bool multi = FUSED && a.k > 0;
    if (!multi) {   // small launches of the tuned windows (a lone frame, two, three): their own copies of the kernels + the HALF tail, nlm_small.hip
        bool handled = false;
        const int rc = nlm_dispatch_small(ctx, p, a, s, FMT, FUSED, &handled);
        if (handled) return rc;
    }
    if (p->search_lo == -10 && p->search_hi == 11 && p->patch_lo == -3 && p->patch_hi == 4) {   // 21x21 / 7x7 (benchmark)
        if (multi) return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, FUSED>(ctx, a, s);
        return launch_strip<-10, 11, -3, 4, 8, 4, FMT, FUSED, false>(ctx, a, s);
    }
    if (p->search_lo == -7 && p->search_hi == 7 && p->patch_lo == -3 && p->patch_hi == 3) {     // nonlocal.comp:5-6 as shipped
        if (multi) return launch_strip<-7, 7, -3, 3, 8, 4, FMT, FUSED, FUSED>(ctx, a, s);
        return launch_strip<-7, 7, -3, 3, 8, 4, FMT, FUSED, false>(ctx, a, s);
    }
    // Any other search window: the same strip kernel with the search range -- and the texel format -- as run-time arguments
    // (LDS pitch no longer a folded constant: a few % slower), instantiated in nlm_rt.hip (patches up to 9x9) and nlm_rt4.hip
    // (10x10 .. 16x16).
    a.slo = p->search_lo; a.shi = p->search_hi; a.fmt = FMT;
    {
        bool handled = false;
        int rc = nlm_dispatch_rt8(ctx, p, a, s, FUSED, &handled);
        if (handled) return rc;
        rc = nlm_dispatch_rt4(ctx, p, a, s, FUSED, &handled);
        if (handled) return rc;
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
    // Every output frame of a launch is computed concurrently from the input frames around it: an output that IS one of the sequence's
    // frames (filtering in place, ping-pong tables shifted by a slot) would be overwritten while other workgroups still read it, and a
    // buffer given twice would be written by two frames.  Refused here, like mid_bilateral_batch does (the library-internal callers of
    // nlm_temporal_out -- the frame pipeline, the sharded call -- own their rings and slots).
    if (frames && out && n_frames >= 1 && count >= 1 && first >= 0 && first + count <= n_frames) {
        std::unordered_set<const void *> inputs(frames, frames + n_frames), outputs;
        for (int t = 0; t < count; ++t) {
            if (!out[t]) continue;                                                   // (reported as NULL below)
            MID_REQUIRE(!inputs.count((const void *)out[t]),
                        "nlm_temporal: out[%d] is also a frame of the sequence (in-place / aliased filtering is not supported)", t);
            MID_REQUIRE(outputs.insert((const void *)out[t]).second, "nlm_temporal: out[%d] appears twice", t);
        }
    }
    return mid::nlm_temporal_out(ctx, p, frames, n_frames, k, first, count, (void *const *)out, 0, stream);
}

int mid::nlm_temporal_out(mid_ctx *ctx, const mid_nlm_params *p, const void *const *frames, int n_frames, int k,
                          int first, int count, void *const *out, int out_u8, void *stream, int corunning)
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
        a.n_frames = hi - lo + 1; a.k = k; a.first = c0 - lo; a.count = cn; a.out_u8 = out_u8; a.corunning = corunning;
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
