// recording_host.cpp -- TEST RESOURCE (tests/test_gpu_z_recording.py, tools/recording_replay.sh): a compiled host of the recording
// API (no HIP header: the C-ABI only).
//   recording_host <w> <h> <frames> <repetitions> <out.raw> [ldr]
// ldr: instead of the NLM sequence, the PNG path of a single-image mode per frame -- mid_unpack_u8, mid_bilateral (r = 4), mid_pack_u8 --
// for all <frames> frames: 3 x frames short launches (out.raw then holds the last frame's RGBA8 result decoded again).
//  Records COLD -- the process's first launch of every kernel is
// inside the recording -- the reference's literal multi-frame sequence, submits it, repeats the sequence call by call, compares the
// bytes, and times both (per sequence, the stream drained once per loop).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mi_denoise.h"
#define CK(x) do { if ((x) != MID_OK) { fprintf(stderr, "%s: %s\n", #x, mid_last_error()); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int w = atoi(argv[1]), h = atoi(argv[2]), n = atoi(argv[3]), reps = atoi(argv[4]);
    const size_t px = (size_t)w * h;
    mid_ctx *ctx; CK(mid_ctx_create(0, &ctx));
    std::vector<void *> fr(n);
    std::vector<float> host(px * 4);
    void *hp; CK(mid_alloc_host(ctx, px * 16, &hp));
    for (int i = 0; i < n; ++i) {
        unsigned s = 12345u + i;
        for (size_t j = 0; j < px * 4; ++j) { s = s * 1664525u + 1013904223u; ((float *)hp)[j] = (j & 3) == 3 ? 1.0f : (s >> 8) * (1.0f / 16777216.0f); }
        CK(mid_alloc(ctx, px * 16, &fr[i]));
        CK(mid_memcpy_h2d(ctx, fr[i], hp, px * 16, nullptr)); CK(mid_stream_sync(ctx, nullptr));
    }
    void *W, *out; CK(mid_alloc(ctx, px * 32, &W)); CK(mid_alloc(ctx, px * 16, &out));
    mid_nlm_params p; memset(&p, 0, sizeof p);
    p.width = w; p.height = h; p.filteringParameter = 0.5f; p.search_lo = -7; p.search_hi = 7; p.patch_lo = -3; p.patch_hi = 3; p.format = MID_FMT_RGBA32F;
    mid_normalize_params pn = {w, h};
    const bool ldr = argc > 6 && !strcmp(argv[6], "ldr");
    mid_bilateral_params bp; memset(&bp, 0, sizeof bp);
    bp.width = w; bp.height = h; bp.spatialSigma = 2.0f; bp.colorSigma = 0.2f; bp.radius = 4; bp.layout = MID_LAYOUT_TEXTURE; bp.format = MID_FMT_RGBA32F;
    auto sequence = [&]() -> int {
        if (ldr) {      // (the frames' first px*4 bytes read as RGBA8 texels; W as the float image, out as the filtered one, W again as the u8 result)
            for (int i = 0; i < n; ++i) {
                CK(mid_unpack_u8(ctx, (const uint8_t *)fr[i], px * 4, 0, (float *)W, nullptr));
                CK(mid_bilateral(ctx, &bp, W, (mid_pixel *)out, nullptr));
                CK(mid_pack_u8(ctx, (const float *)out, px * 4, (uint8_t *)W + px * 16, nullptr));
            }
            CK(mid_unpack_u8(ctx, (const uint8_t *)W + px * 16, px * 4, 0, (float *)out, nullptr));     // (the last frame's u8 result, as floats, where the comparison reads)
            return 0;
        }
        CK(mid_memset(ctx, W, 0, px * 32, nullptr));
        for (int i = 0; i < n; ++i) CK(mid_nlm_accum(ctx, &p, fr[n / 2], fr[i], (mid_weightinfo *)W, nullptr));
        CK(mid_normalize(ctx, &pn, (const mid_weightinfo *)W, (mid_pixel *)out, nullptr));
        return 0;
    };
    mid_recording *rec;
    CK(mid_record_begin(ctx, nullptr));
    if (sequence()) return 1;
    CK(mid_record_end(ctx, nullptr, &rec));
    int nodes, kernels; CK(mid_recording_info(rec, &nodes, &kernels));
    std::vector<float> a(px * 4), b(px * 4);
    CK(mid_recording_submit(rec, nullptr));
    CK(mid_memcpy_d2h(ctx, a.data(), out, px * 16, nullptr));       // (pageable: outside the recording this is fine)
    CK(mid_memset(ctx, out, 0, px * 16, nullptr));
    if (sequence()) return 1;
    CK(mid_memcpy_d2h(ctx, b.data(), out, px * 16, nullptr));
    const bool same = memcmp(a.data(), b.data(), px * 16) == 0;
    double t[2];
    for (int m = 0; m < 2; ++m) {
        double best = 1e30;
        for (int pass = 0; pass < 5; ++pass) {
            CK(mid_stream_sync(ctx, nullptr));
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) { if (m == 0) { if (sequence()) return 1; } else CK(mid_recording_submit(rec, nullptr)); }
            CK(mid_stream_sync(ctx, nullptr));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
            if (ms < best) best = ms;
        }
        t[m] = best;
    }
    // ldr only: the library's own answer to many short launches -- ONE launch over all frames (mid_bilateral_batch) against one mid_bilateral per frame
    double tb[2] = {0, 0};
    if (ldr) {
        std::vector<void *> outs(n);
        std::vector<const void *> ins(fr.begin(), fr.end());
        for (int i = 0; i < n; ++i) CK(mid_alloc(ctx, px * 16, &outs[i]));
        for (int m = 0; m < 2; ++m) {
            double best = 1e30;
            for (int pass = 0; pass < 5; ++pass) {
                CK(mid_stream_sync(ctx, nullptr));
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r) {
                    if (m == 0) { for (int i = 0; i < n; ++i) CK(mid_bilateral(ctx, &bp, fr[i], (mid_pixel *)outs[i], nullptr)); }
                    else CK(mid_bilateral_batch(ctx, &bp, ins.data(), (mid_pixel *const *)outs.data(), n, nullptr));
                }
                CK(mid_stream_sync(ctx, nullptr));
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
                if (ms < best) best = ms;
            }
            tb[m] = best;
        }
        for (int i = 0; i < n; ++i) CK(mid_free(ctx, outs[i]));
    }
    FILE *f = fopen(argv[5], "wb"); fwrite(a.data(), 1, px * 16, f); fclose(f);
    printf("{\"sequence\": \"%s\", \"w\": %d, \"h\": %d, \"dispatches\": %d, \"nodes\": %d, \"kernels\": %d, \"same\": %s, \"by_call_ms\": %.4f, \"submit_ms\": %.4f",
           ldr ? "ldr_bilateral_r4" : "nlm_literal", w, h, ldr ? 3 * n + 1 : n + 1, nodes, kernels, same ? "true" : "false", t[0], t[1]);
    if (ldr) printf(", \"bilateral_one_launch_per_frame_ms\": %.4f, \"bilateral_batch_one_launch_ms\": %.4f", tb[0], tb[1]);
    printf("}\n");
    CK(mid_recording_destroy(rec));
    mid_ctx_destroy(ctx);
    return same ? 0 : 3;
}
