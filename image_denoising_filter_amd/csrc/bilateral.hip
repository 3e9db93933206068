// bilateral.hip -- bilateral filters for gfx950 (replaces shaders/bialteral.comp,
// bialteral_linear.comp and bialteral_layers.comp).
//
// Design.  A workgroup stages its output tile plus a `radius` halo in LDS as float4 texels
// (coalesced 16 B/lane HBM reads, halo zero-filled per the reference's out-of-image policy).
// A wave owns 64 adjacent columns x P rows: lane l owns one column and P vertically adjacent
// outputs, so each LDS texel it reads (one conflict-free ds_read_b128) feeds up to P taps.
// Per tap: 3 sub, 3 mul/fma (colour distance), 1 fma folding the colour and the spatial
// exponent, 1 v_exp_f32, 4 fma + 1 add.  The two exp() of the shader become one:
//     exp(-.5 (sd/ss)^2) * exp(-.5 (cd/sc)^2) = exp2(ks*(i^2+j^2) + kc*cd^2)
// with ks = -.5*log2(e)/ss^2, kc = -.5*log2(e)/sc^2; the spatial term is wave-uniform.
//
// The texture and linear variants differ only in how the tile is addressed when it is filled
// (2-D zero border vs flat index with row wrap-around), exactly the difference between
// bialteral.comp:58-59 and bialteral_linear.comp:58.
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mid {

struct BilArgs {
    int w, h;
    float ks, kc;          // exponent scales (log2 domain)
    float sc, inv_sc;      // sqrt(-kc) and its reciprocal: the tiled kernels carry the range scale in the guide colours
    int tiles_x, tiles_y;
    const void *in;
    float4 *out;           // plain / fused-layers output
    mid_weightinfo *W;     // layers accumulate mode
    int n_layers;
    const uint32_t *layers[16];
};

// Frame tables of the batched plain bilateral (mid_bilateral_batch): passed by value in kernarg space like the NLM
// kernels' tables.  Single-frame launches pass the empty BilOne instead, so their kernarg block stays small.
struct BilBatch { FrameTable in; OutTable out; };
struct BilOne {};

__device__ __forceinline__ unsigned xcd_remap_b(unsigned bid, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u, i = bid >> 3;
    return x * q + (x < r ? x : r) + i;
}

// MODE 0: plain bilateral (range weight and colour from `in`)
// MODE 1: layers, accumulate one layer into W      (one dispatch of bialteral_layers.comp)
// MODE 2: layers, all layers fused + normalize     (loop src/main.cpp:1610-1623 + normalize.comp)
template <int R, int P, int NW, int FMT, bool LINEAR, int MODE, typename BT>
__global__ __launch_bounds__(NW * 64) void bilateral_kernel(const BilArgs a, const BT bt)
{
    constexpr bool BATCH = std::is_same<BT, BilBatch>::value;
    constexpr int TILE_W = 64, TILE_H = NW * P;
    constexpr int LW = TILE_W + 2 * R, LH = TILE_H + 2 * R;
    constexpr int MR = P + 2 * R;   // tile rows a lane walks per column offset

    extern __shared__ float4 lds[];
    float4 *img_t = lds;                                  // colour source
    float4 *gde_t = (MODE == 0) ? lds : lds + LW * LH;    // range-weight source (guide)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned flat;
    const void *in = a.in;
    float4 *out = a.out;
    if constexpr (BATCH) {
        // frames in launch order, tiles remapped inside their frame (every XCD gets a contiguous run of each frame)
        const unsigned tiles = (unsigned)(a.tiles_x * a.tiles_y);
        const unsigned fz = blockIdx.x / tiles;
        flat = xcd_remap_in_frame(blockIdx.x - fz * tiles, tiles, fz);
        in = bt.in.p[fz];
        out = (float4 *)bt.out.p[fz];
    } else {
        flat = xcd_remap_b(blockIdx.x, gridDim.x);
    }
    const int ty = (int)(flat / (unsigned)a.tiles_x), tx = (int)(flat - (unsigned)ty * a.tiles_x);
    const int w = a.w, h = a.h;
    const int X0 = tx * TILE_W, Y0 = ty * TILE_H;
    const int gx = X0 + lane, yb = Y0 + wv * P;
    const bool wave_active = yb < h;

    // Register-staged double buffer for the guide tile (MODE 2): the NEXT layer's RGBA8 texels -- PF u32 per thread -- are
    // requested from HBM before the current layer's tap loop and only decoded into LDS after it, between the two barriers
    // that already separate the passes (the first layer's request goes out ahead of the image tile's fill).  Same texels,
    // same decode, same scale as fill_tile ==> the same bits in LDS; not a byte more LDS.  Measured: 4 layers at r = 8
    // 0.702 -> 0.689 ms, r = 4 0.207 -> 0.203, r = 10 1.083 -> 1.061, outputs bit-identical (LABNOTES R5.2).
    constexpr int PF = (LW * LH + NW * 64 - 1) / (NW * 64);
    uint32_t pf[PF];
    auto prefetch = [&](const uint32_t *layer) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int t = tid + j * NW * 64;
            const int ty_ = t / LW, tx_ = t - ty_ * LW;
            const int x = X0 - R + tx_, y = Y0 - R + ty_;
            pf[j] = 0u;                                    // decode_rgba8(0) == vec4(0): the out-of-image texel
            if (t < LW * LH && (unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) pf[j] = layer[(size_t)y * w + x];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int t = tid + j * NW * 64;
            const float4 v = decode_rgba8(pf[j]);
            if (t < LW * LH) gde_t[t] = make_float4(v.x * a.sc, v.y * a.sc, v.z * a.sc, v.w);
        }
    };
    if (MODE == 2 && a.n_layers > 0) prefetch(a.layers[0]);      // (no layers: the loop below does not run, the output is the magenta sentinel)
    // The guide colours (the image itself in MODE 0) are pre-multiplied by sqrt(-kc), so -|dc|^2 is already the
    // colour part of the exp2 argument and the FMA chain can start from the spatial term: 12 ops per tap, not 13.
    bool mine = true;                                     // every texel THIS thread stored in the colour tile has alpha == 1.0f
    fill_tile<FMT, LINEAR>(img_t, LW, LH, in, w, h, X0 - R, Y0 - R, tid, NW * 64, MODE == 0 ? a.sc : 1.0f, &mine);

    // Is every texel of the colour tile opaque (alpha == 1.0f, the usual case away from the image border, where out-of-image
    // texels are vec4(0))?  Then the alpha accumulator repeats the weight accumulator operation for operation --
    // fma(1.0f, wt, acc.w) and accw + wt round alike, both start at 0 -- and the tap loop need not carry it: one FMA of 13
    // instructions per tap less, the same bits.  Round 6: r = 8 0.1541 -> 0.1490 ms (-3.3 %), r = 10 / 20 -3 %, r = 4 -1 %, every
    // output's sha equal; 4 layers fused at r = 8 0.6443 -> 0.6225 ms (-3.4 %), r = 10 -3.7 %, r = 4 -2 %
    // (profiles/r06_ab_bilateral_opaque_alpha.txt, LABNOTES R6.7).
    // How the workgroup agrees.  Plain bilateral: __syncthreads_and (4 B of static LDS: the 40 KB tile then fits three times per
    // CU instead of four, which this kernel does not notice -- 0.1547-0.1553 against 0.1552-0.1554 ms).  Layer modes: their two
    // tiles are EXACTLY half the CU's LDS, and one static word would halve the occupancy (+20 %, measured); the guide tile is
    // still unused at this point, so its first word carries the vote.
    bool alpha_one = false;
    if constexpr (MODE == 0) {
        alpha_one = __syncthreads_and(mine) != 0;
    } else {
        unsigned *vote = (unsigned *)gde_t;
        if (tid == 0) vote[0] = 1u;
        __syncthreads();
        if (!mine) vote[0] = 0u;
        __syncthreads();
        alpha_one = __builtin_amdgcn_readfirstlane((int)vote[0]) != 0;      // (the pass loop's first barrier comes before the guide tile is written)
    }

    // spatial exponent by |j|: ks * j^2 (wave-uniform)
    float sj[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) sj[j] = a.ks * (float)(j * j);

    float4 tot[P];
    float totw[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { tot[k] = make_float4(0.f, 0.f, 0.f, 0.f); totw[k] = 0.f; }

    const int n_pass = (MODE == 2) ? a.n_layers : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        if (MODE == 2) {
            __syncthreads();                               // every wave has left the previous pass's tap loop
            commit();
            if (pass + 1 < n_pass) prefetch(a.layers[pass + 1]);
        } else if (MODE != 0) {
            __syncthreads();
            fill_tile<MID_FMT_RGBA8, false>(gde_t, LW, LH, a.layers[pass], w, h, X0 - R, Y0 - R, tid, NW * 64, a.sc);
        }
        __syncthreads();
        if (!wave_active) continue;

        float cr[P], cg[P], cb[P];   // centre guide colour (texColor / layerColor)
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float4 c = gde_t[(wv * P + k + R) * LW + lane + R];
            cr[k] = c.x; cg[k] = c.y; cb[k] = c.z;
        }
        float4 acc[P];
        float accw[P];
#pragma unroll
        for (int k = 0; k < P; ++k) { acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); accw[k] = 0.f; }

        auto taps = [&](auto a1_tag) {
        constexpr bool A1 = decltype(a1_tag)::value;
        for (int i = -R; i <= R; ++i) {
            const float si = a.ks * (float)(i * i);
            float sij[R + 1];
#pragma unroll
            for (int j = 0; j <= R; ++j) sij[j] = si + sj[j];
            const int base = (wv * P) * LW + lane + R + i;
            // Rows in groups of RG = 2 (measured against 0/3/4/6/9/18: profiles/r03_ab_bilateral_exp_bursts.txt): exponent arguments
            // of the group (plain VALU), then ALL its v_exp_f32 in one burst at raised issue priority, then the accumulates, still
            // at raised priority.  A transcendental mixed into other waves' plain instructions costs far more than its own 8
            // cycles (tools/microbench10/11.hip: 8 exps + 88 FMAs 357 cycles per group per SIMD against 67 + 211 alone; 274
            // with s_setprio around the exp burst); scheduling barriers keep the three phases apart.
            // Same instructions per tap, same accumulation order: identical output bits.
            // (Each group's two ds_read_b128 are followed directly by their first use.  Issuing them one group ahead was measured in
            // round 6 and buys nothing -- 8 waves per SIMD cover the LDS latency already; profiles/r06_ab_bilateral_readahead.txt.)
            constexpr int RG = 2;
#pragma unroll
            for (int m0 = 0; m0 < MR; m0 += RG) {
                float4 cc[RG];
                float ar[RG][P];
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int m = m0 + r;
                    if (m >= MR) continue;
                    const float4 g = gde_t[base + m * LW];
                    cc[r] = g;
                    if (MODE != 0) cc[r] = img_t[base + m * LW];
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        const int j = m - R - k;
                        if (j < -R || j > R) continue;
                        const float dx = cr[k] - g.x, dy = cg[k] - g.y, dz = cb[k] - g.z;
                        ar[r][k] = fmaf(-dz, dz, fmaf(-dy, dy, fmaf(-dx, dx, sij[j < 0 ? -j : j])));
                    }
                    if (MODE != 0) asm volatile("" ::"v"(g.w), "v"(accw[P - 1]));
                    // opaque form: the colour texel's alpha is not used any more; kept formally live so that the tile read stays a
                    // ds_read_b128 (4 LDS cycles; the ds_read_b96 the compiler would shrink it to takes 8: LDS utilisation 0.45 against 0.22)
                    if constexpr (A1) asm volatile("" ::"v"(cc[r].w));
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int r = 0; r < RG; ++r)
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        const int j = m0 + r - R - k;
                        if (m0 + r >= MR || j < -R || j > R) continue;
                        ar[r][k] = __builtin_amdgcn_exp2f(ar[r][k]);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RG; ++r)
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        const int j = m0 + r - R - k;
                        if (m0 + r >= MR || j < -R || j > R) continue;
                        const float wt = ar[r][k];
                        const float4 c = cc[r];
                        acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);
                        acc[k].z = fmaf(c.z, wt, acc[k].z);
                        if constexpr (!A1) acc[k].w = fmaf(c.w, wt, acc[k].w);
                        accw[k] += wt;
                    }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (A1) {
#pragma unroll
            for (int k = 0; k < P; ++k) acc[k].w = accw[k];
        }
        };
        if (alpha_one) taps(std::true_type{}); else taps(std::false_type{});
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (MODE == 0) { acc[k].x *= a.inv_sc; acc[k].y *= a.inv_sc; acc[k].z *= a.inv_sc; }   // back to unscaled colours
            tot[k].x += acc[k].x; tot[k].y += acc[k].y; tot[k].z += acc[k].z; tot[k].w += acc[k].w;
            totw[k] += accw[k];
        }
    }

    if (!wave_active || gx >= w) return;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int gy = yb + k;
        if (gy >= h) break;
        const size_t idx = (size_t)gy * w + gx;
        if (MODE == 0) {
            out[idx] = make_float4(tot[k].x / totw[k], tot[k].y / totw[k], tot[k].z / totw[k], tot[k].w / totw[k]);
        } else if (MODE == 2) {
            float4 o;
            if (totw[k] == 0.0f) o = make_float4(1.f, 0.f, 1.f, 1.f);
            else o = make_float4(tot[k].x / totw[k], tot[k].y / totw[k], tot[k].z / totw[k], tot[k].w / totw[k]);
            out[idx] = o;
        } else {
            float4 *wp = (float4 *)(a.W + idx);
            float4 wc = wp[0], nw = wp[1];
            wc.x += tot[k].x; wc.y += tot[k].y; wc.z += tot[k].z; wc.w += tot[k].w;
            nw.x += totw[k];
            wp[0] = wc; wp[1] = nw;
        }
    }
}

// Any radius without a tuned instantiation: the same LDS-tiled scheme with the radius as a run-time
// value (loops not unrolled, spatial exponent computed per tap row).  8 waves x 2 rows per workgroup.
template <int FMT, bool LINEAR, int MODE, typename BT>
__global__ __launch_bounds__(512) void bilateral_rt_kernel(const BilArgs a, const int R, const BT bt)
{
    constexpr bool BATCH = std::is_same<BT, BilBatch>::value;
    constexpr int NW = 8, P = 2, TILE_W = 64, TILE_H = NW * P;
    const int LW = TILE_W + 2 * R, LH = TILE_H + 2 * R;
    extern __shared__ float4 lds[];
    float4 *img_t = lds;
    float4 *gde_t = (MODE == 0) ? lds : lds + LW * LH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned flat;
    const void *in = a.in;
    float4 *out = a.out;
    if constexpr (BATCH) {
        const unsigned tiles = (unsigned)(a.tiles_x * a.tiles_y);
        const unsigned fz = blockIdx.x / tiles;
        flat = xcd_remap_in_frame(blockIdx.x - fz * tiles, tiles, fz);
        in = bt.in.p[fz];
        out = (float4 *)bt.out.p[fz];
    } else {
        flat = xcd_remap_b(blockIdx.x, gridDim.x);
    }
    const int ty = (int)(flat / (unsigned)a.tiles_x), tx = (int)(flat - (unsigned)ty * a.tiles_x);
    const int w = a.w, h = a.h;
    const int X0 = tx * TILE_W, Y0 = ty * TILE_H;
    const int gx = X0 + lane, yb = Y0 + wv * P;
    const bool wave_active = yb < h;

    fill_tile<FMT, LINEAR>(img_t, LW, LH, in, w, h, X0 - R, Y0 - R, tid, NW * 64, MODE == 0 ? a.sc : 1.0f);
    float4 tot[P];
    float totw[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { tot[k] = make_float4(0.f, 0.f, 0.f, 0.f); totw[k] = 0.f; }
    const int n_pass = (MODE == 2) ? a.n_layers : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        if (MODE != 0) {
            __syncthreads();
            fill_tile<MID_FMT_RGBA8, false>(gde_t, LW, LH, a.layers[pass], w, h, X0 - R, Y0 - R, tid, NW * 64, a.sc);
        }
        __syncthreads();
        if (!wave_active) continue;
        float cr[P], cg[P], cb[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float4 c = gde_t[(wv * P + k + R) * LW + lane + R];
            cr[k] = c.x; cg[k] = c.y; cb[k] = c.z;
        }
        float4 acc[P];
        float accw[P];
#pragma unroll
        for (int k = 0; k < P; ++k) { acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); accw[k] = 0.f; }
        for (int i = -R; i <= R; ++i) {
            const float si = a.ks * (float)(i * i);
            const int base = (wv * P) * LW + lane + R + i;
            // Tile row m feeds output k = 0 with row offset j = m - R and output k = 1 with j = m - R - 1; the first row has only the
            // k = 0 tap, the last only the k = 1 tap, and the 2R rows between them go in PAIRS with their four exps as one burst at
            // raised issue priority, like the tuned kernel (per output the taps are still added in row order: same bits).
            auto arg_of = [&](const float4 &g, int k, int j) {
                const float dx = cr[k] - g.x, dy = cg[k] - g.y, dz = cb[k] - g.z;
                return fmaf(-dz, dz, fmaf(-dy, dy, fmaf(-dx, dx, fmaf(a.ks, (float)(j * j), si))));
            };
            auto add_tap = [&](const float4 &c, int k, float wt) {
                acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);
                acc[k].z = fmaf(c.z, wt, acc[k].z); acc[k].w = fmaf(c.w, wt, acc[k].w);
                accw[k] += wt;
            };
            {
                const float4 g = gde_t[base];
                add_tap(MODE != 0 ? img_t[base] : g, 0, exp2_hw(arg_of(g, 0, -R)));
            }
            for (int m = 1; m < 2 * R; m += 2) {
                const float4 g0 = gde_t[base + m * LW], g1 = gde_t[base + (m + 1) * LW];
                const float4 c0 = MODE != 0 ? img_t[base + m * LW] : g0, c1 = MODE != 0 ? img_t[base + (m + 1) * LW] : g1;
                float w00 = arg_of(g0, 0, m - R), w01 = arg_of(g0, 1, m - R - 1), w10 = arg_of(g1, 0, m + 1 - R), w11 = arg_of(g1, 1, m - R);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                w00 = __builtin_amdgcn_exp2f(w00); w01 = __builtin_amdgcn_exp2f(w01);
                w10 = __builtin_amdgcn_exp2f(w10); w11 = __builtin_amdgcn_exp2f(w11);
                __builtin_amdgcn_sched_barrier(0);
                add_tap(c0, 0, w00); add_tap(c0, 1, w01); add_tap(c1, 0, w10); add_tap(c1, 1, w11);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const int m = 2 * R + 1;
                const float4 g = gde_t[base + m * LW];
                add_tap(MODE != 0 ? img_t[base + m * LW] : g, 1, exp2_hw(arg_of(g, 1, R)));
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (MODE == 0) { acc[k].x *= a.inv_sc; acc[k].y *= a.inv_sc; acc[k].z *= a.inv_sc; }
            tot[k].x += acc[k].x; tot[k].y += acc[k].y; tot[k].z += acc[k].z; tot[k].w += acc[k].w;
            totw[k] += accw[k];
        }
    }
    if (!wave_active || gx >= w) return;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int gy = yb + k;
        if (gy >= h) break;
        const size_t idx = (size_t)gy * w + gx;
        if (MODE == 1) {
            float4 *wp = (float4 *)(a.W + idx);
            float4 wc = wp[0], nw = wp[1];
            wc.x += tot[k].x; wc.y += tot[k].y; wc.z += tot[k].z; wc.w += tot[k].w;
            nw.x += totw[k];
            wp[0] = wc; wp[1] = nw;
        } else {
            float4 o;
            if (MODE == 2 && totw[k] == 0.0f) o = make_float4(1.f, 0.f, 1.f, 1.f);
            else o = make_float4(tot[k].x / totw[k], tot[k].y / totw[k], tot[k].z / totw[k], tot[k].w / totw[k]);
            out[idx] = o;
        }
    }
}

// Last resort (tile does not fit LDS): one thread per pixel, global fetches.
template <int FMT, bool LINEAR, int MODE>
__global__ __launch_bounds__(256) void bilateral_generic_kernel(const BilArgs a, int R)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= a.w || y >= a.h) return;
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    float totw = 0.f;
    const int n_pass = (MODE == 2) ? a.n_layers : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        float4 ctr;
        if (MODE != 0) ctr = fetch_texture<MID_FMT_RGBA8>(a.layers[pass], a.w, a.h, x, y);
        else ctr = LINEAR ? fetch_linear<FMT>(a.in, a.w, a.h, x, y) : fetch_texture<FMT>(a.in, a.w, a.h, x, y);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float accw = 0.f;
        for (int j = -R; j <= R; ++j)
            for (int i = -R; i <= R; ++i) {
                float4 g, c;
                if (MODE != 0) {
                    g = fetch_texture<MID_FMT_RGBA8>(a.layers[pass], a.w, a.h, x + i, y + j);
                    c = fetch_texture<FMT>(a.in, a.w, a.h, x + i, y + j);
                } else {
                    g = LINEAR ? fetch_linear<FMT>(a.in, a.w, a.h, x + i, y + j) : fetch_texture<FMT>(a.in, a.w, a.h, x + i, y + j);
                    c = g;
                }
                const float dx = ctr.x - g.x, dy = ctr.y - g.y, dz = ctr.z - g.z;
                const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                const float wt = exp2_hw(fmaf(d2, a.kc, a.ks * (float)(i * i + j * j)));
                acc.x = fmaf(c.x, wt, acc.x); acc.y = fmaf(c.y, wt, acc.y);
                acc.z = fmaf(c.z, wt, acc.z); acc.w = fmaf(c.w, wt, acc.w);
                accw += wt;
            }
        tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w;
        totw += accw;
    }
    const size_t idx = (size_t)y * a.w + x;
    if (MODE == 1) {
        float4 *wp = (float4 *)(a.W + idx);
        float4 wc = wp[0], nw = wp[1];
        wc.x += tot.x; wc.y += tot.y; wc.z += tot.z; wc.w += tot.w;
        nw.x += totw;
        wp[0] = wc; wp[1] = nw;
    } else {
        float4 o;
        if (MODE == 2 && totw == 0.0f) o = make_float4(1.f, 0.f, 1.f, 1.f);
        else o = make_float4(tot.x / totw, tot.y / totw, tot.z / totw, tot.w / totw);
        a.out[idx] = o;
    }
}

template <int R, int P, int NW, int FMT, bool LINEAR, int MODE, typename BT>
static int launch_tiled(mid_ctx *ctx, BilArgs &a, const BT &bt, int n_frames, hipStream_t s)
{
    constexpr int LW = 64 + 2 * R, LH = NW * P + 2 * R;
    constexpr size_t lds_bytes = (size_t)LW * LH * sizeof(float4) * (MODE == 0 ? 1 : 2);
    auto kern = bilateral_kernel<R, P, NW, FMT, LINEAR, MODE, BT>;
    if ((int)lds_bytes > ctx->lds_max)
        return set_error(MID_ERR_UNSUPPORTED, "bilateral tile needs %zu B of LDS, device offers %d", lds_bytes, ctx->lds_max);
    if (int rc = ensure_lds(ctx, (const void *)kern, lds_bytes)) return rc;
    a.tiles_x = (int)cdiv(a.w, 64);
    a.tiles_y = (int)cdiv(a.h, NW * P);
    hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles_x * a.tiles_y * (unsigned)n_frames), dim3(NW * 64), lds_bytes, s, a, bt);
    MID_HIP(hipGetLastError());
    return MID_OK;
}

// n_frames > 1 only with BT = BilBatch (plain bilateral, MODE 0): grid = tiles x frames.
template <int FMT, bool LINEAR, int MODE, typename BT = BilOne>
static int dispatch_radius(mid_ctx *ctx, int radius, BilArgs &a, hipStream_t s, const BT &bt = BT{}, int n_frames = 1)
{
    // Tile shapes by A/B on MI355X (tools/ab_bil.py): the kernel is latency-sensitive, so many
    // independent waves (P = 2 rows per lane, 8 waves per workgroup) beat deeper register blocking.
    switch (radius) {
    case 4:  return launch_tiled<4, 2, 8, FMT, LINEAR, MODE, BT>(ctx, a, bt, n_frames, s);    // BASELINE config 1 window
    case 8:  return launch_tiled<8, 2, 8, FMT, LINEAR, MODE, BT>(ctx, a, bt, n_frames, s);    // BASELINE configs[1] and [3] (layer modes: also best of six shapes, profiles/r05_ab_layer_tile_shapes.txt)
    case 10: return launch_tiled<10, 2, 16, FMT, LINEAR, MODE, BT>(ctx, a, bt, n_frames, s);  // CPU path window, src/main.cpp:1819
    case 20:                                                                 // TEXEL_WINDOW as shipped
        return launch_tiled<20, 1, 8, FMT, LINEAR, MODE, BT>(ctx, a, bt, n_frames, s);          // 80 KB tile: two workgroups per CU (or image + guide tile)
    default: break;
    }
    {   // run-time radius, LDS tiled
        const size_t lds_bytes = (size_t)(64 + 2 * radius) * (16 + 2 * radius) * sizeof(float4) * (MODE == 0 ? 1 : 2);
        if ((int)lds_bytes <= ctx->lds_max) {
            auto kern = bilateral_rt_kernel<FMT, LINEAR, MODE, BT>;
            if (int rc = ensure_lds(ctx, (const void *)kern, (size_t)ctx->lds_max)) return rc;
            a.tiles_x = (int)cdiv(a.w, 64);
            a.tiles_y = (int)cdiv(a.h, 16);
            hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles_x * a.tiles_y * (unsigned)n_frames), dim3(512), lds_bytes, s, a, radius, bt);
            MID_HIP(hipGetLastError());
            return MID_OK;
        }
    }
    // (unreachable for the plain bilateral: its single tile fits LDS for every legal radius; only the two-tile
    // layer modes at r > 16 get here, and those are never batched)
    if (n_frames != 1) return set_error(MID_ERR_UNSUPPORTED, "bilateral: no batched kernel for radius %d", radius);
    dim3 grid(cdiv(a.w, 16), cdiv(a.h, 16));
    hipLaunchKernelGGL((bilateral_generic_kernel<FMT, LINEAR, MODE>), grid, dim3(256), 0, s, a, radius);
    MID_HIP(hipGetLastError());
    return MID_OK;
}

static int check_params(const mid_bilateral_params *p, const char *who)
{
    MID_REQUIRE(p != nullptr, "%s: params is NULL", who);
    MID_REQUIRE(p->width > 0 && p->height > 0, "%s: bad size %dx%d", who, p->width, p->height);
    MID_REQUIRE((long)p->width * p->height < (1l << 30), "%s: image too large", who);
    MID_REQUIRE(p->spatialSigma > 0.f && p->colorSigma > 0.f, "%s: sigmas must be > 0", who);
    MID_REQUIRE(p->radius >= 1 && p->radius <= 24, "%s: radius %d outside 1..24", who, p->radius);
    MID_REQUIRE(p->format == MID_FMT_RGBA32F || p->format == MID_FMT_RGBA8, "%s: unknown format %d", who, p->format);
    MID_REQUIRE(p->layout == MID_LAYOUT_TEXTURE || p->layout == MID_LAYOUT_LINEAR, "%s: unknown layout %d", who, p->layout);
    return MID_OK;
}

static void fill_scales(const mid_bilateral_params *p, BilArgs &a)
{
    a.w = p->width; a.h = p->height;
    a.ks = (float)(-0.5 * 1.4426950408889634 / ((double)p->spatialSigma * (double)p->spatialSigma));
    a.kc = (float)(-0.5 * 1.4426950408889634 / ((double)p->colorSigma * (double)p->colorSigma));
    a.sc = (float)(sqrt(0.5 * 1.4426950408889634) / (double)p->colorSigma);
    a.inv_sc = (float)(1.0 / (double)a.sc);
}

}  // namespace mid

using namespace mid;

extern "C" int mid_bilateral(mid_ctx *ctx, const mid_bilateral_params *p, const void *in,
                             mid_pixel *out, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = check_params(p, "bilateral")) return rc;
    MID_REQUIRE(in && out, "bilateral: NULL image pointer");
    MID_REQUIRE((const void *)in != (const void *)out, "bilateral: in-place filtering is not supported");
    BilArgs a{};
    fill_scales(p, a);
    a.in = in; a.out = (float4 *)out;
    const bool lin = p->layout == MID_LAYOUT_LINEAR, u8 = p->format == MID_FMT_RGBA8;
    if (lin) return u8 ? dispatch_radius<MID_FMT_RGBA8, true, 0>(ctx, p->radius, a, b.s)
                       : dispatch_radius<MID_FMT_RGBA32F, true, 0>(ctx, p->radius, a, b.s);
    return u8 ? dispatch_radius<MID_FMT_RGBA8, false, 0>(ctx, p->radius, a, b.s)
              : dispatch_radius<MID_FMT_RGBA32F, false, 0>(ctx, p->radius, a, b.s);
}

extern "C" int mid_bilateral_layers_accum(mid_ctx *ctx, const mid_bilateral_params *p, const void *in,
                                          const uint32_t *layer, mid_weightinfo *W, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = check_params(p, "bilateral_layers_accum")) return rc;
    MID_REQUIRE(in && layer && W, "bilateral_layers_accum: NULL pointer");
    // NLM/layers are only ever bound to textures in the reference (src/main.cpp:1406-1428).
    MID_REQUIRE(p->layout == MID_LAYOUT_TEXTURE, "bilateral_layers_accum: layers exist for the texture layout only");
    BilArgs a{};
    fill_scales(p, a);
    a.in = in; a.W = W; a.n_layers = 1; a.layers[0] = layer;
    return p->format == MID_FMT_RGBA8 ? dispatch_radius<MID_FMT_RGBA8, false, 1>(ctx, p->radius, a, b.s)
                                      : dispatch_radius<MID_FMT_RGBA32F, false, 1>(ctx, p->radius, a, b.s);
}

extern "C" int mid_bilateral_layers(mid_ctx *ctx, const mid_bilateral_params *p, const void *in,
                                    const uint32_t *const *layers, int n_layers, mid_pixel *out, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = check_params(p, "bilateral_layers")) return rc;
    MID_REQUIRE(in && layers && out, "bilateral_layers: NULL pointer");
    MID_REQUIRE((const void *)out != in, "bilateral_layers: out is the input image (in-place filtering is not supported)");
    MID_REQUIRE(p->layout == MID_LAYOUT_TEXTURE, "bilateral_layers: layers exist for the texture layout only");
    MID_REQUIRE(n_layers >= 0 && n_layers <= 16, "bilateral_layers: n_layers %d outside 0..16", n_layers);
    BilArgs a{};
    fill_scales(p, a);
    a.in = in; a.out = (float4 *)out; a.n_layers = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        MID_REQUIRE(layers[i] != nullptr, "bilateral_layers: layer %d is NULL", i);
        a.layers[i] = layers[i];
    }
    return p->format == MID_FMT_RGBA8 ? dispatch_radius<MID_FMT_RGBA8, false, 2>(ctx, p->radius, a, b.s)
                                      : dispatch_radius<MID_FMT_RGBA32F, false, 2>(ctx, p->radius, a, b.s);
}

extern "C" int mid_bilateral_batch(mid_ctx *ctx, const mid_bilateral_params *p, const void *const *in,
                                   mid_pixel *const *out, int n_frames, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = check_params(p, "bilateral_batch")) return rc;
    MID_REQUIRE(in && out, "bilateral_batch: NULL table");
    MID_REQUIRE(n_frames >= 1, "bilateral_batch: n_frames %d < 1", n_frames);
    {
        // All frames of a launch run concurrently: an output that is ANY frame's input (not only its own) would be
        // written while other workgroups still read it -- ping-pong tables shifted by one slot, say.  Reject it.
        std::unordered_set<const void *> inputs;
        for (int i = 0; i < n_frames; ++i) {
            MID_REQUIRE(in[i] && out[i], "bilateral_batch: frame %d is NULL", i);
            inputs.insert(in[i]);
        }
        std::unordered_set<const void *> outputs;
        for (int i = 0; i < n_frames; ++i) {
            MID_REQUIRE(!inputs.count((const void *)out[i]),
                        "bilateral_batch: out[%d] is also an input frame of this call (in-place / aliased filtering is not supported)", i);
            MID_REQUIRE(outputs.insert((const void *)out[i]).second, "bilateral_batch: out[%d] appears twice", i);
        }
    }
    const bool lin = p->layout == MID_LAYOUT_LINEAR, u8 = p->format == MID_FMT_RGBA8;
    for (int c0 = 0; c0 < n_frames; c0 += kMaxFrames) {          // one launch per kMaxFrames frames
        const int cn = n_frames - c0 < kMaxFrames ? n_frames - c0 : kMaxFrames;
        BilArgs a{};
        fill_scales(p, a);
        BilBatch bt{};
        for (int i = 0; i < cn; ++i) { bt.in.p[i] = in[c0 + i]; bt.out.p[i] = out[c0 + i]; }
        int rc;
        if (lin) rc = u8 ? dispatch_radius<MID_FMT_RGBA8, true, 0, BilBatch>(ctx, p->radius, a, b.s, bt, cn)
                         : dispatch_radius<MID_FMT_RGBA32F, true, 0, BilBatch>(ctx, p->radius, a, b.s, bt, cn);
        else rc = u8 ? dispatch_radius<MID_FMT_RGBA8, false, 0, BilBatch>(ctx, p->radius, a, b.s, bt, cn)
                     : dispatch_radius<MID_FMT_RGBA32F, false, 0, BilBatch>(ctx, p->radius, a, b.s, bt, cn);
        if (rc) return rc;
    }
    return MID_OK;
}
