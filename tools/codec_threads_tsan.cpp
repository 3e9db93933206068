// codec_threads_tsan.cpp -- ThreadSanitizer run of the codecs used the way mi_denoise --animation uses them (round 6): several
// host threads encode and decode different images AT THE SAME TIME, each with its own per-thread limit on the codec's internal
// parallelism (codec::thread_cap, the state behind mid_image_threads).  Every thread must get its own image back, and the file
// bytes must not depend on the limit.  Built and run by tests/test_codecs.py::test_concurrent_codec_calls_under_tsan with
// -fsanitize=thread (CPU build only).
#include <cstdio>
#include <random>
#include <thread>

#include "../image_denoising_filter_amd/csrc/codec/image_io.hpp"

using namespace mid::codec;

int main()
{
    const int NT = 4, W = 700, H = 260;                    // 728 KB of RGBA8 (one deflate segment + filter rows), 2.9 MB of float (17 ZIP chunks)
    std::vector<std::vector<uint8_t>> png_ref(NT), exr_ref(NT);
    std::vector<std::vector<uint8_t>> px8(NT);
    std::vector<std::vector<float>> pxf(NT);
    for (int t = 0; t < NT; ++t) {
        std::mt19937 rng(100 + t);
        px8[t].resize((size_t)W * H * 4);
        pxf[t].resize((size_t)W * H * 4);
        for (auto &v : px8[t]) v = (uint8_t)(rng() >> 24);
        for (auto &v : pxf[t]) v = (float)(rng() >> 8) / 4096.0f;
        std::string e;
        thread_cap() = 1;                                   // reference bytes: serial
        if (!png_encode(px8[t].data(), W, H, png_ref[t], e) || !exr_encode(pxf[t].data(), W, H, exr_ref[t], e)) { printf("encode failed: %s\n", e.c_str()); return 1; }
    }
    thread_cap() = 0;
    std::vector<int> bad(NT, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < NT; ++t)
        th.emplace_back([&, t] {
            thread_cap() = t;                               // 0 = default (up to 16 inner threads), 1, 2, 3
            for (int rep = 0; rep < 3; ++rep) {
                std::vector<uint8_t> f8, ff, d8;
                std::vector<float> df;
                std::string e;
                int w = 0, h = 0;
                if (!png_encode(px8[t].data(), W, H, f8, e) || f8 != png_ref[t]) { bad[t] = 1; return; }
                if (!png_decode(f8, w, h, d8, e) || w != W || h != H || d8 != px8[t]) { bad[t] = 2; return; }
                if (!exr_encode(pxf[t].data(), W, H, ff, e) || ff != exr_ref[t]) { bad[t] = 3; return; }
                if (!exr_decode(ff, w, h, df, e) || w != W || h != H || df != pxf[t]) { bad[t] = 4; return; }
            }
            if (thread_cap() != t) bad[t] = 5;              // nobody else touched this thread's setting
        });
    for (auto &t : th) t.join();
    for (int t = 0; t < NT; ++t) if (bad[t]) { printf("thread %d failed at step %d\n", t, bad[t]); return 1; }
    if (thread_cap() != 0) { printf("the main thread's setting changed\n"); return 1; }
    printf("concurrent codec calls done\n");
    return 0;
}
