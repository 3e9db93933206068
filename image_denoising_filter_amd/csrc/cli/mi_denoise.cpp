// mi_denoise -- drop-in command-line driver (replaces the reference's `vulkan_denoice`,
// src/main.cpp:1935-1994) on top of libmi_denoise.so.
//
//   mi_denoise [image] [options]
//
// With no options it behaves like the reference: the positional argument is the target image
// (default Animations/CornellBox/Animation01_LDR_0000.png, src/main.cpp:1945); six GPU modes and two
// CPU runs execute in the reference's order (src/main.cpp:1952-1985), print the same banners and
// timing lines (:1924-1933) and write output{-linear|-nonlinear}{-nlm|-bialteral}[-multiframe]
// [-overlap][-layers].{png|exr} and output-cpu.{png|exr} into the current directory (:1677-1686,
// :1870-1901).  Unlike the reference it does not need to be started from the repo root (kernels are
// linked in, not loaded from shaders/*.spv).  The parameters the reference hard-codes (windows
// shaders/*.comp:5-6, sigmas/h src/main.cpp:806,870,875,1833-1835, CPU window :1819, device :1321)
// are options whose defaults are those values.
//
// There is no CPU fallback for the GPU modes: if the device is missing they fail and the exit code
// is EXIT_FAILURE.  The two "Running on CPU" modes are the reference's own RunOnCPU feature
// (src/main.cpp:1732-1921), implemented here as part of the driver.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <condition_variable>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <string>
#include <atomic>
#include <system_error>
#include <thread>
#include <vector>

#include "../../../include/mi_denoise.h"

namespace fs = std::filesystem;

// console colours of the timing lines, src/main.cpp:21-23
#define FOREGROUND_COLOR "\033[38;2;0;0;0m"
#define BACKGROUND_COLOR "\033[48;2;0;255;0m"
#define CLEAR_COLOR      "\033[0m"

struct Pixel { float r, g, b, a; };   // src/main.cpp:39-41

struct Options {
    std::string image = "Animations/CornellBox/Animation01_LDR_0000.png";   // src/main.cpp:1945
    std::string outdir = ".";
    int device = 0;                 // deviceId{0}, src/main.cpp:1321
    int radius = 20;                // TEXEL_WINDOW, shaders/bialteral.comp:5
    float sigma_s = 2.0f, sigma_c = 0.2f;      // src/main.cpp:806,875
    float nlm_h = 0.5f;             // src/main.cpp:870
    int search_lo = -7, search_hi = 7, patch_lo = -3, patch_hi = 3;   // shaders/nonlocal.comp:5-6, half-open
    int temporal_k = -1;            // <0: the reference's frame list; >=0: window t-k..t+k of the sorted sequence
    int cpu_radius = 10;            // windowSize, src/main.cpp:1819
    float cpu_sigma_s = 10.0f, cpu_sigma_c = 0.2f;   // src/main.cpp:1833-1835
    bool cpu_blue_bug = true;       // pow(texColor.b - texColor.b, 2), src/main.cpp:1850
    std::vector<int> cpu_threads = {1, 8};     // src/main.cpp:1979,1984
    bool run_gpu = true, run_cpu = true;
    std::string modes = "all";      // comma list of: bilateral,layers,linear,nlm,multiframe,overlap
    bool animation = false;         // new capability: temporal NLM of EVERY frame of the sequence
    int gpus = 1;                   // animation mode: frame blocks over this many devices
    bool share_device = false;      // animation mode: every block on --device (rehearsal of --gpus N on fewer devices)
    bool halo_rccl = false;         // animation mode: blocks resident in HBM, halo frames GPU to GPU over RCCL (mid_nlm_temporal_sharded)
    bool pageable_host = false;     // reference modes: keep decoded images and results in ordinary memory (the C-ABI bounces them)
    long pinned_mb = 16384;         // animation mode: at most this much page-locked host memory (inputs + outputs); the rest is pageable
    int io_threads = 0;             // animation mode: files decoded / encoded at a time (0 = min(16, hardware threads))
};

#define MID_CHECK(call)                                                                          \
    do {                                                                                         \
        if ((call) != 0) throw std::runtime_error(std::string(#call) + ": " + mid_last_error()); \
    } while (0)

// ---- RunOnCPU, src/main.cpp:1732-1921 -----------------------------------------------------------------
// The reference's CPU bilateral: window `radius`, double-precision pow/sqrt/exp stored to float, float
// accumulators, rows [radius, h-radius] and columns [radius, w-radius] INCLUSIVE (the last row/column
// read one past the image; here flat indices >= N read a zero pixel), range distance over r,g only when
// blue_bug (the reference's typo), alpha forced to 1, untouched border = Pixel{}.
static void cpu_bilateral_refpath(const std::vector<Pixel> &in, int w, int h, int radius, float spatialSigma,
                                  float colorSigma, bool blue_bug, int numThreads, std::vector<Pixel> &out)
{
    const long n = (long)w * h;
    out.assign((size_t)n, Pixel{0.f, 0.f, 0.f, 0.f});
    auto fetch = [&](long idx) -> Pixel { return (idx >= 0 && idx < n) ? in[(size_t)idx] : Pixel{0.f, 0.f, 0.f, 0.f}; };
    for (int y = radius; y <= h - radius && y < h; ++y) {
#pragma omp parallel for default(shared) num_threads(numThreads)
        for (int x = radius; x <= w - radius; ++x) {
            if (x >= w) continue;
            const Pixel texColor = in[(size_t)y * w + x];
            float normWeight = 0.0f, wr = 0.f, wg = 0.f, wb = 0.f;
            for (int i = -radius; i <= radius; ++i)
                for (int j = -radius; j <= radius; ++j) {
                    const float spatialDistance = (float)std::sqrt((double)(float)std::pow((double)i, 2.0) + std::pow((double)j, 2.0));
                    const float spatialWeight = (float)std::exp(-0.5 * std::pow((double)(spatialDistance / spatialSigma), 2.0));
                    const Pixel cur = fetch((long)w * (i + y) + j + x);
                    const double db = blue_bug ? std::pow((double)(texColor.b - texColor.b), 2.0) : std::pow((double)(texColor.b - cur.b), 2.0);
                    const float colorDistance = (float)std::sqrt(std::pow((double)(texColor.r - cur.r), 2.0) + std::pow((double)(texColor.g - cur.g), 2.0) + db);
                    const float colorWeight = (float)std::exp(-0.5 * std::pow((double)(colorDistance / colorSigma), 2.0));
                    const float resultWeight = spatialWeight * colorWeight;
                    wr += cur.r * resultWeight; wg += cur.g * resultWeight; wb += cur.b * resultWeight;
                    normWeight += resultWeight;
                }
            out[(size_t)y * w + x] = Pixel{wr / normWeight, wg / normWeight, wb / normWeight, 1.0f};
        }
    }
}

// A named ROCTx range (mid_range_push/pop): shows up in `rocprofv3 --marker-trace`, does nothing otherwise.  The reference
// brackets the same stages with timestamp queries (src/main.cpp:793-796,812-814,842-844).
struct TraceRange {
    explicit TraceRange(const std::string &name) { (void)mid_range_push(name.c_str()); }
    ~TraceRange() { (void)mid_range_pop(); }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

class DenoiseApplication {
    Options opt;
    double m_execMs = 0, m_transferMs = 0;

    static bool is_hdr(const std::string &p) { return fs::path(p).extension() == ".exr"; }   // src/main.cpp:1380

    // A decoded image.  With a context it lives in page-locked memory (mid_image_load_pinned), the HIP counterpart of the
    // host-visible staging buffer the reference memcpy's its decoded pixels into (LoadImageDataToBuffer,
    // src/main.cpp:1105-1142), so every copy the GPU modes issue is a plain asynchronous DMA.  Without one (RunOnCPU), or
    // when page-locked memory runs out, it is ordinary memory, which the C-ABI moves through its own pinned bounce buffers.
    struct HostImage {
        int w = 0, h = 0, format = 0;
        mid_ctx *pin_ctx = nullptr;          // non-NULL: `img.data` is pinned and freed through this context
        mid_image img{};
        HostImage() = default;
        HostImage(const HostImage &) = delete;
        HostImage &operator=(const HostImage &) = delete;
        HostImage(HostImage &&o) noexcept : w(o.w), h(o.h), format(o.format), pin_ctx(o.pin_ctx), img(o.img) { o.img.data = nullptr; }
        HostImage &operator=(HostImage &&o) noexcept
        {
            if (this != &o) { release(); w = o.w; h = o.h; format = o.format; pin_ctx = o.pin_ctx; img = o.img; o.img.data = nullptr; }
            return *this;
        }
        void release()
        {
            if (!img.data) return;
            if (pin_ctx) (void)mid_image_free_pinned(pin_ctx, &img); else mid_image_free(&img);
            img.data = nullptr;
        }
        ~HostImage() { release(); }
        size_t size() const { return (size_t)w * h * (format == MID_FMT_RGBA32F ? 16 : 4); }
        const uint8_t *data() const { return (const uint8_t *)img.data; }
    };

    static HostImage load(const std::string &path, bool force_png, mid_ctx *ctx = nullptr)
    {
        // layers are always decoded as PNG (src/main.cpp:1396 passes a_isHDR=false)
        if (force_png && is_hdr(path)) throw std::runtime_error("layer " + path + " is not a PNG");
        HostImage h;
        if (ctx && mid_image_load_pinned(ctx, path.c_str(), &h.img) == MID_OK) h.pin_ctx = ctx;
        else if (mid_image_load(path.c_str(), &h.img))
            throw std::runtime_error(mid_last_error());        // lodepng error -> runtime_error, src/main.cpp:202
        h.w = h.img.width; h.h = h.img.height; h.format = h.img.format;
        return h;
    }

    std::string out_path(const std::string &name) const { return (fs::path(opt.outdir) / name).string(); }

    // Files of a sequence are decoded / encoded CONCURRENTLY, one file per worker thread: the reference does its image I/O on one
    // thread (lodepng / tinyexr calls, src/main.cpp:155,196,1699,1717), and a PNG's inflate is one serial stream, so files taken
    // one after the other leave all but one core idle for most of the time.  fn(i) runs for i in [first, n) on `files_at_a_time()`
    // threads; each of them lets its codec calls use the host threads that are left (mid_image_threads: all of them with
    // --io-threads 1, one with a file per host thread).  The first exception ends the loop and is rethrown here.
    int host_threads() const { return (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())); }
    int files_at_a_time(int n) const { return std::max(1, std::min(opt.io_threads > 0 ? opt.io_threads : host_threads(), n)); }
    template <class Fn> void for_each_file(int first, int n, Fn fn) const
    {
        if (n <= first) return;
        const int nt = files_at_a_time(n - first), codec_threads = std::max(1, host_threads() / nt);
        std::atomic<int> next{first};
        std::vector<std::string> err(nt);
        auto worker = [&](int t) {
            const int before = mid_image_threads(codec_threads);
            try { for (int i = next++; i < n; i = next++) fn(i); }
            catch (const std::exception &e) { err[t] = e.what(); next = n; }
            (void)mid_image_threads(before);
        };
        std::vector<std::thread> th;
        try { for (int t = 1; t < nt; ++t) th.emplace_back(worker, t); }
        catch (const std::system_error &) {}              // no more threads to be had: the ones that started (and this one) do the work
        worker(0);                                         // the calling thread is worker 0
        for (auto &t : th) t.join();
        for (auto &e : err) if (!e.empty()) throw std::runtime_error(e);
    }

public:
    explicit DenoiseApplication(const Options &o) : opt(o) {}
    double GetTransferMs() const { return m_transferMs; }
    double GetExecMs() const { return m_execMs; }

    // Frame / layer discovery, src/main.cpp:1343-1378.  Differences, all documented in DESIGN.md:
    // entries are sorted by name (the reference uses directory order, which is unspecified), and the
    // 4-character frame id comes from the file name's stem (the reference's find(".") breaks on "./x").
    void discover(std::vector<std::string> &frames, std::vector<std::string> &layers, bool want_frames, bool want_layers) const
    {
        const fs::path target(opt.image);
        fs::path parent = target.parent_path();
        if (parent.empty()) parent = ".";
        const std::string stem = target.stem().string();
        const std::string imageID = stem.size() >= 4 ? stem.substr(stem.size() - 4) : stem;
        std::vector<fs::path> entries;
        for (auto &p : fs::directory_iterator(parent)) entries.push_back(p.path());
        std::sort(entries.begin(), entries.end());
        for (auto &e : entries) {
            if (fs::is_directory(e)) {
                if (!want_layers) continue;
                std::vector<fs::path> inner;
                for (auto &pp : fs::directory_iterator(e)) inner.push_back(pp.path());
                std::sort(inner.begin(), inner.end());
                for (auto &l : inner)
                    if (!fs::is_directory(l) && l.string().find(imageID) != std::string::npos) layers.push_back(l.string());
            } else if (e.extension() == target.extension() && want_frames) {
                frames.push_back(e.string());
            }
        }
    }

    // RunOnGPU, src/main.cpp:1307-1730
    void RunOnGPU(bool nlmFilter, bool nonlinear, bool multiframe, bool execAndCopyOverlap, bool useLayers)
    {
        const bool linear = !nonlinear;                                              // :1311
        TraceRange mode_range(std::string("RunOnGPU ") + (linear ? "linear" : "nonlinear") + (nlmFilter ? " nlm" : " bialteral") +
                              (multiframe ? " multiframe" : "") + (execAndCopyOverlap ? " overlap" : "") + (useLayers ? " layers" : ""));
        if (!(nlmFilter || !multiframe)) throw std::runtime_error("multiframe works only with nlm");        // assert :1315
        if (!(multiframe || !execAndCopyOverlap)) throw std::runtime_error("overlap needs multiframe");     // assert :1316
        if (linear && (nlmFilter || useLayers)) throw std::runtime_error("nlm/layers exist for the texture path only");
        m_execMs = m_transferMs = 0;
        std::cout << "\tinit hip for device " << opt.device << "\n";
        mid_ctx *ctx = nullptr;
        MID_CHECK(mid_ctx_create(opt.device, &ctx));
        struct CtxGuard { mid_ctx *c; ~CtxGuard() { mid_ctx_destroy(c); } } guard{ctx};

        std::cout << "\tloading image data\n";
        std::vector<std::string> frameNames, layerNames;
        discover(frameNames, layerNames, multiframe, useLayers);
        const bool hdr = is_hdr(opt.image);
        mid_ctx *pin = opt.pageable_host ? nullptr : ctx;              // where decoded images live
        if (opt.pageable_host) std::cout << "\thost buffers: pageable (bounced inside the library)\n";
        HostImage target = [&] { TraceRange r("decode target"); return load(opt.image, false, pin); }();
        const int w = target.w, h = target.h;
        const size_t npix = (size_t)w * h, out_bytes = npix * sizeof(Pixel);
        const int fmt = target.format;
        auto check_dims = [&](const HostImage &im, const std::string &name) {
            if (im.w != w || im.h != h || im.format != fmt) throw std::runtime_error(name + ": size/format differs from the target image");
        };

        // the read-back target: page-locked like the reference's staging buffer (GetImageFromGPU maps it, src/main.cpp:91-96);
        // ordinary memory if that allocation fails
        struct Result {
            mid_ctx *c; void *pinned = nullptr; std::vector<Pixel> fallback;
            Result(mid_ctx *ctx, size_t n, bool want_pinned) : c(ctx)
            {
                if (!want_pinned || mid_alloc_host(c, n * sizeof(Pixel), &pinned)) { pinned = nullptr; fallback.resize(n); }
            }
            ~Result() { if (pinned) (void)mid_free_host(c, pinned); }
            Pixel *data() { return pinned ? (Pixel *)pinned : fallback.data(); }
        } result(ctx, npix, !opt.pageable_host);
        mid_timer *tm = nullptr;
        MID_CHECK(mid_timer_create(ctx, &tm));
        struct TimerGuard { mid_timer *t; ~TimerGuard() { mid_timer_destroy(t); } } tguard{tm};
        auto timed = [&](double &acc, auto &&fn) {
            MID_CHECK(mid_timer_tick(tm, nullptr));
            fn();
            MID_CHECK(mid_timer_tock(tm, nullptr));
            float ms = 0;
            MID_CHECK(mid_timer_ms(tm, &ms));
            acc += ms;
        };

        std::cout << "\tperforming computations\n";
        if (nlmFilter && multiframe) {
            // frame list: the reference loads the target first and then every sibling with the same
            // extension (the target again among them), src/main.cpp:1390-1393; in overlap mode it uploads 10
            // frames and dispatches over the first 9 (:1341,1539-1573).  --temporal-k switches to the explicit
            // window t-k..t+k of the sorted sequence instead.
            std::vector<std::string> list;
            if (opt.temporal_k >= 0) {
                auto it = std::find_if(frameNames.begin(), frameNames.end(), [&](const std::string &f) { return fs::equivalent(f, opt.image); });
                const int t = it == frameNames.end() ? 0 : (int)(it - frameNames.begin());
                for (int f = std::max(0, t - opt.temporal_k); f <= std::min((int)frameNames.size() - 1, t + opt.temporal_k); ++f) list.push_back(frameNames[f]);
                if (list.empty()) list.push_back(opt.image);
            } else {
                list.push_back(opt.image);
                list.insert(list.end(), frameNames.begin(), frameNames.end());
                if (execAndCopyOverlap && list.size() > 9) list.resize(9);
            }
            std::vector<HostImage> frames(list.size());
            for_each_file(0, (int)list.size(), [&](int i) {
                frames[i] = load(list[i], false, pin);
                check_dims(frames[i], list[i]);
            });
            std::vector<const void *> ptrs;
            for (auto &f : frames) ptrs.push_back(f.data());
            mid_nlm_params p{w, h, opt.nlm_h, opt.search_lo, opt.search_hi, opt.patch_lo, opt.patch_hi, fmt};
            float t[3] = {0, 0, 0};
            MID_CHECK(mid_nlm_multiframe(ctx, &p, target.data(), ptrs.data(), (int)ptrs.size(), (mid_pixel *)result.data(),
                                         execAndCopyOverlap ? 1 : 0, t));
            m_execMs = t[1]; m_transferMs = t[2];
        } else {
            void *dIn = nullptr, *dOut = nullptr;
            MID_CHECK(mid_alloc(ctx, target.size(), &dIn));
            MID_CHECK(mid_alloc(ctx, out_bytes, &dOut));
            { TraceRange r("upload target"); timed(m_transferMs, [&] { MID_CHECK(mid_memcpy_h2d(ctx, dIn, target.data(), target.size(), nullptr)); }); }
            TraceRange dispatch_range("dispatch");
            if (nlmFilter) {                                                                    // single-frame NLM, :1577-1606 with one frame
                mid_nlm_params p{w, h, opt.nlm_h, opt.search_lo, opt.search_hi, opt.patch_lo, opt.patch_hi, fmt};
                const void *fr[1] = {dIn};
                mid_pixel *ou[1] = {(mid_pixel *)dOut};
                timed(m_execMs, [&] { MID_CHECK(mid_nlm_temporal(ctx, &p, fr, 1, 0, 0, 1, ou, nullptr)); });
            } else if (useLayers) {                                                             // :1608-1623 + normalize
                std::vector<void *> dLayers;
                std::vector<HostImage> layerImgs(layerNames.size());
                for_each_file(0, (int)layerNames.size(), [&](int i) {
                    layerImgs[i] = load(layerNames[i], true, pin);
                });
                for (size_t li = 0; li < layerNames.size(); ++li) {
                    const std::string &ln = layerNames[li];
                    std::cout << "\t\tfeeding layer to texture\n";
                    HostImage &l = layerImgs[li];
                    if (l.w != w || l.h != h) throw std::runtime_error(ln + ": layer size differs from the target image");
                    void *d = nullptr;
                    MID_CHECK(mid_alloc(ctx, l.size(), &d));
                    dLayers.push_back(d);
                    timed(m_transferMs, [&] { MID_CHECK(mid_memcpy_h2d(ctx, d, l.data(), l.size(), nullptr)); });
                    MID_CHECK(mid_stream_sync(ctx, nullptr));             // (the decoded layers are released with `layerImgs`)
                }
                mid_bilateral_params p{w, h, opt.sigma_s, opt.sigma_c, opt.radius, MID_LAYOUT_TEXTURE, fmt};
                timed(m_execMs, [&] {
                    MID_CHECK(mid_bilateral_layers(ctx, &p, dIn, (const uint32_t *const *)dLayers.data(), (int)dLayers.size(), (mid_pixel *)dOut, nullptr));
                });
                for (void *d : dLayers) mid_free(ctx, d);
            } else {                                                                            // plain bialteral, :1654-1659
                mid_bilateral_params p{w, h, opt.sigma_s, opt.sigma_c, opt.radius, linear ? MID_LAYOUT_LINEAR : MID_LAYOUT_TEXTURE, fmt};
                timed(m_execMs, [&] { MID_CHECK(mid_bilateral(ctx, &p, dIn, (mid_pixel *)dOut, nullptr)); });
            }
            std::cout << "\tgetting image back\n";
            (void)mid_range_pop();                                                              // "dispatch" ends, "download" begins
            (void)mid_range_push("download");
            timed(m_transferMs, [&] { MID_CHECK(mid_memcpy_d2h(ctx, result.data(), dOut, out_bytes, nullptr)); });
            MID_CHECK(mid_stream_sync(ctx, nullptr));
            mid_free(ctx, dIn); mid_free(ctx, dOut);
        }

        std::string outputFileName{"output"};                                                   // :1677-1686
        outputFileName += linear ? "-linear" : "-nonlinear";
        outputFileName += nlmFilter ? "-nlm" : "-bialteral";
        outputFileName += multiframe ? "-multiframe" : "";
        outputFileName += execAndCopyOverlap ? "-overlap" : "";
        outputFileName += useLayers ? "-layers" : "";
        { TraceRange r("encode + write " + outputFileName); save(outputFileName, result.data(), w, h, hdr); }
        std::cout << "\tcleaning up\n";
    }

    // Animation mode (BASELINE config 5; not in the reference, which filters one target per run): every
    // sibling frame of the target is denoised with temporal NLM over frames t-k..t+k.  Frames are split
    // into contiguous blocks, one per device; each device streams its block plus k halo frames on either
    // side through the 3-stream pipeline (mid_sequence_nlm_range): all frames sit in host memory, so by default
    // the halo is simply uploaded twice and needs no device-to-device exchange.  With --halo rccl every block is
    // uploaded once, stays resident in its GPU's HBM, and the halo frames travel GPU to GPU over RCCL/xGMI
    // (mid_nlm_temporal_sharded, csrc/sharded.cpp).  Outputs: output-animation-<frame file name>.
    void RunAnimation()
    {
        std::vector<std::string> frameNames, layerNames;
        discover(frameNames, layerNames, true, false);
        if (frameNames.empty()) throw std::runtime_error("no frames next to " + opt.image);
        const int n = (int)frameNames.size(), k = opt.temporal_k < 0 ? 2 : opt.temporal_k;
        std::cout << "\tloading " << n << " frames\n";
        // Frames are decoded STRAIGHT INTO pinned host memory (mid_image_load_pinned) and the results land in pinned
        // buffers too, so every copy of the pipeline is a true asynchronous DMA -- the reference memcpy's its decoded
        // pixels into mapped staging memory the same way (LoadImageDataToBuffer, src/main.cpp:1105-1142).  Measured,
        // 16 x 1080p RGBA32F through the pipeline: pinned 12.8 ms, pageable vectors 20.2 ms, vectors registered in
        // place 25.1 ms.  Page-locking is per process, so the one context used for loading serves every device.
        mid_ctx *io = nullptr;
        MID_CHECK(mid_ctx_create(opt.device, &io));
        // Page-locked memory is a bounded resource: at most --pinned-mb of it (default 16 GiB, inputs and outputs
        // together) is requested, and a frame whose pinned allocation fails -- or that would exceed the budget -- is
        // decoded into ordinary pageable memory instead (the pipeline accepts both; HIP then stages that copy and its
        // overlap is lost, nothing else changes).  A long or high-resolution sequence degrades, it does not abort.
        struct Pinned {                                   // owns the frames and outputs (pinned or pageable); released before `io`
            mid_ctx *c;
            std::vector<mid_image> frames;
            std::vector<char> frame_pinned;
            std::vector<void *> outs;
            std::vector<char> out_pinned;
            ~Pinned()
            {
                for (size_t i = 0; i < frames.size(); ++i) {
                    if (frame_pinned[i]) (void)mid_image_free_pinned(c, &frames[i]);
                    else mid_image_free(&frames[i]);
                }
                for (size_t i = 0; i < outs.size(); ++i) {
                    if (out_pinned[i]) (void)mid_free_host(c, outs[i]);
                    else free(outs[i]);
                }
                mid_ctx_destroy(c);
            }
        } pin{io, {}, {}, {}, {}};
        const size_t pinned_budget = (size_t)std::max(0l, opt.pinned_mb) << 20;
        size_t pinned_bytes = 0, frame_bytes_guess = 0;
        int n_pageable = 0;
        const auto tl0 = std::chrono::steady_clock::now();
        // Frame 0 is decoded first (its size fixes how many frames fit the page-locked budget: inputs and outputs share it frame
        // by frame, input i is pinned only if output i can be as well); the others are decoded concurrently (for_each_file).
        pin.frames.assign(n, mid_image{});
        pin.frame_pinned.assign(n, 0);
        auto decode = [&](int i, bool pinned) {            // throws on a bad file; falls back to pageable memory when no pinned memory is left
            mid_image img{};
            if (pinned && mid_image_load_pinned(io, frameNames[i].c_str(), &img)) pinned = false;
            if (!pinned && mid_image_load(frameNames[i].c_str(), &img)) throw std::runtime_error(mid_last_error());
            pin.frames[i] = img;
            pin.frame_pinned[i] = pinned ? 1 : 0;
        };
        decode(0, pinned_budget > 0);                       // (the first frame's size is not known yet: any non-zero budget admits it)
        frame_bytes_guess = (size_t)pin.frames[0].width * pin.frames[0].height * (pin.frames[0].format == MID_FMT_RGBA32F ? 16 : 4);
        const size_t n_pin = pinned_budget / (2 * frame_bytes_guess);     // frames whose input AND output fit the budget
        const int io_threads = files_at_a_time(n);
        for_each_file(1, n, [&](int i) {
            decode(i, (size_t)i < n_pin);
            const mid_image &a = pin.frames[i], &b = pin.frames[0];
            if (a.width != b.width || a.height != b.height || a.format != b.format)
                throw std::runtime_error(frameNames[i] + ": size/format differs from the first frame");
        });
        for (int i = 0; i < n; ++i) { if (pin.frame_pinned[i]) pinned_bytes += 2 * frame_bytes_guess; else ++n_pageable; }
        const double load_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
        const int w = pin.frames[0].width, h = pin.frames[0].height, fmt = pin.frames[0].format;
        // LDR frames come back as RGBA8: the read-back conversion of GetImageFromGPU (:97-103) runs on the device
        // (mid_sequence_nlm_range_u8), a quarter of the download
        const bool hdr = fmt == MID_FMT_RGBA32F;
        const size_t out_bytes = (size_t)w * h * (hdr ? 16 : 4), in_bytes = out_bytes;
        std::vector<const void *> in(n);
        for (int i = 0; i < n; ++i) {
            in[i] = pin.frames[i].data;
            void *o = nullptr;
            bool pinned = pin.frame_pinned[i] != 0;
            if (pinned && mid_alloc_host(io, out_bytes, &o)) { pinned = false; o = nullptr; }
            if (!pinned) {
                o = malloc(out_bytes);
                if (!o) throw std::runtime_error("out of host memory for the output frames");
                if (pin.frame_pinned[i]) ++n_pageable;
            }
            pin.outs.push_back(o);
            pin.out_pinned.push_back(pinned ? 1 : 0);
        }
        if (n_pageable)
            std::cout << "	" << n_pageable << " frame(s) beyond the page-locked budget (--pinned-mb " << opt.pinned_mb
                      << ") use pageable host memory: their copies are staged by HIP and do not overlap\n";
        const int G = std::max(1, std::min(opt.gpus, n));
        // One context per device, created -- and its code object, streams and allocator warmed by filtering two tiny
        // frames -- BEFORE the clock starts: the timed region below is the frame pipeline itself (uploads, kernels,
        // downloads), like the reference's timestamps bracket its submits and not vkCreateDevice.
        std::vector<mid_ctx *> ctxs(G, nullptr);
        struct CtxGuard { std::vector<mid_ctx *> &v; ~CtxGuard() { for (auto c : v) if (c) mid_ctx_destroy(c); } } guard{ctxs};
        const mid_nlm_params p{w, h, opt.nlm_h, opt.search_lo, opt.search_hi, opt.patch_lo, opt.patch_hi, fmt};
        const auto tw0 = std::chrono::steady_clock::now();
        for (int g = 0; g < G; ++g) {
            MID_CHECK(mid_ctx_create(opt.share_device ? opt.device : opt.device + g, &ctxs[g]));
            // (a) the kernel's code object and the pipeline's streams: two tiny frames through the same entry point
            const int ww = 64, wh = 32;
            mid_nlm_params wp = p;
            wp.width = ww; wp.height = wh;
            std::vector<unsigned char> a((size_t)ww * wh * 16, 0), o((size_t)ww * wh * 16);
            const void *wi[2] = {a.data(), a.data()};
            if (hdr) { mid_pixel *wo[2] = {(mid_pixel *)o.data(), (mid_pixel *)o.data()}; MID_CHECK(mid_sequence_nlm_range(ctxs[g], &wp, wi, 2, k > 0 ? 1 : 0, 0, 1, wo, 1, nullptr)); }
            else { uint8_t *wo[2] = {o.data(), o.data()}; MID_CHECK(mid_sequence_nlm_range_u8(ctxs[g], &wp, wi, 2, k > 0 ? 1 : 0, 0, 1, wo, 1, nullptr)); }
            // (b) the pinned-memory DMA path in both directions at the real frame size (its first use in a process
            // costs several ms): frame 0 up into a scratch buffer, and back down into the first result buffer
            void *scratch = nullptr;
            MID_CHECK(mid_alloc(ctxs[g], std::max(in_bytes, out_bytes), &scratch));
            int rc = mid_memcpy_h2d(ctxs[g], scratch, in[0], in_bytes, nullptr);
            if (!rc) rc = mid_memcpy_d2h(ctxs[g], pin.outs[0], scratch, out_bytes, nullptr);
            if (!rc) rc = mid_stream_sync(ctxs[g], nullptr);
            (void)mid_free(ctxs[g], scratch);
            if (rc) throw std::runtime_error(mid_last_error());
        }
        const double warm_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
        std::vector<std::string> errors(G);
        std::vector<float> kern(G, 0.f), copy(G, 0.f);
        std::vector<std::thread> workers;
        // --halo rccl: one communicator rank per device (one process, ncclCommInitAll); created before the clock starts,
        // like the contexts
        std::vector<mid_comm *> comms(G, nullptr);
        struct CommGuard { std::vector<mid_comm *> &v; ~CommGuard() { for (auto c : v) if (c) (void)mid_comm_destroy(c); } } cguard{comms};
        if (opt.halo_rccl) MID_CHECK(mid_comm_create_all(ctxs.data(), G, comms.data()));
        // all ranks report whether their local set-up succeeded; arrive() returns true only if every one of them did
        struct Rendezvous {
            std::mutex m; std::condition_variable cv; int arrived = 0; const int n; bool ok = true;
            explicit Rendezvous(int n_) : n(n_) {}
            bool arrive(bool mine)
            {
                std::unique_lock<std::mutex> l(m);
                ok = ok && mine;
                if (++arrived == n) cv.notify_all();
                else cv.wait(l, [&] { return arrived == n; });
                return ok;
            }
        } rendezvous(G);
        const auto t0 = std::chrono::steady_clock::now();
        for (int g = 0; g < G; ++g)
            workers.emplace_back([&, g] {
                try {
                    const int q = n / G, r = n % G, start = g * q + std::min(g, r), count = q + (g < r ? 1 : 0);
                    if (opt.halo_rccl) {
                        // GPU-resident variant: the block is uploaded once and stays in HBM; the k frames on either side come
                        // from the neighbouring devices over xGMI (ncclSend/ncclRecv, one group) while the interior frames are
                        // being filtered; no frame is uploaded twice.  Every rank calls in, also one that owns no frame.
                        // mid_nlm_temporal_sharded is collective, so the ranks first finish everything that can fail locally
                        // (buffers, uploads, timers, the halo receive buffers) and AGREE that all of them did; if one did not,
                        // no rank enters the exchange.  A failure after that point aborts every communicator, so that ranks
                        // already waiting for the failed one return with an error instead of hanging.
                        mid_ctx *ctx = ctxs[g];
                        struct Dev { mid_ctx *c; std::vector<void *> p; ~Dev() { for (auto q : p) if (q) (void)mid_free(c, q); } } din{ctx, {}}, dout{ctx, {}};
                        struct TG { mid_timer *a = nullptr, *b = nullptr; ~TG() { (void)mid_timer_destroy(a); (void)mid_timer_destroy(b); } } tg;
                        bool ready = false;
                        try {
                            MID_CHECK(mid_timer_create(ctx, &tg.a));
                            MID_CHECK(mid_timer_create(ctx, &tg.b));
                            MID_CHECK(mid_timer_tick(tg.b, nullptr));
                            for (int i = 0; i < count; ++i) {
                                void *d = nullptr;
                                MID_CHECK(mid_alloc(ctx, in_bytes, &d)); din.p.push_back(d);
                                MID_CHECK(mid_memcpy_h2d(ctx, d, in[start + i], in_bytes, nullptr));
                                MID_CHECK(mid_alloc(ctx, (size_t)w * h * 16, &d)); dout.p.push_back(d);
                            }
                            if (!hdr && count) { void *u8 = nullptr; MID_CHECK(mid_alloc(ctx, out_bytes, &u8)); din.p.push_back(u8); }
                            MID_CHECK(mid_comm_reserve(comms[g], in_bytes, k));
                            MID_CHECK(mid_timer_tock(tg.b, nullptr));
                            MID_CHECK(mid_stream_sync(ctx, nullptr));
                            ready = true;
                        } catch (const std::exception &e) { errors[g] = e.what(); }
                        if (!rendezvous.arrive(ready)) return;          // (the rank that failed has recorded why)
                        try {
                            mid_timer *tk = tg.a, *tc = tg.b;
                            MID_CHECK(mid_timer_tick(tk, nullptr));
                            MID_CHECK(mid_nlm_temporal_sharded(comms[g], &p, din.p.data(), n, k, (mid_pixel *const *)dout.p.data(), nullptr));
                            MID_CHECK(mid_timer_tock(tk, nullptr));
                            void *u8 = (!hdr && count) ? din.p.back() : nullptr;
                            for (int i = 0; i < count; ++i) {
                                if (hdr) MID_CHECK(mid_memcpy_d2h(ctx, pin.outs[start + i], dout.p[i], out_bytes, nullptr));
                                else {   // GetImageFromGPU's u8 conversion (:97-103) on the device, then a quarter of the bytes come back
                                    MID_CHECK(mid_pack_u8(ctx, (const float *)dout.p[i], (size_t)w * h * 4, (uint8_t *)u8, nullptr));
                                    MID_CHECK(mid_memcpy_d2h(ctx, pin.outs[start + i], u8, out_bytes, nullptr));
                                }
                            }
                            MID_CHECK(mid_stream_sync(ctx, nullptr));
                            float ms = 0.f;
                            MID_CHECK(mid_timer_ms(tk, &ms)); kern[g] = ms;
                            MID_CHECK(mid_timer_ms(tc, &ms)); copy[g] = ms;
                        } catch (const std::exception &e) {
                            errors[g] = e.what();
                            for (auto c : comms) if (c) (void)mid_comm_abort(c);
                        }
                        return;
                    }
                    if (count == 0) return;
                    mid_ctx *ctx = ctxs[g];
                    float t[3] = {0, 0, 0};
                    if (hdr) {
                        std::vector<mid_pixel *> o(count);
                        for (int i = 0; i < count; ++i) o[i] = (mid_pixel *)pin.outs[start + i];
                        MID_CHECK(mid_sequence_nlm_range(ctx, &p, in.data(), n, k, start, count, o.data(), 1, t));
                    } else {
                        std::vector<uint8_t *> o(count);
                        for (int i = 0; i < count; ++i) o[i] = (uint8_t *)pin.outs[start + i];
                        MID_CHECK(mid_sequence_nlm_range_u8(ctx, &p, in.data(), n, k, start, count, o.data(), 1, t));
                    }
                    kern[g] = t[1]; copy[g] = t[2];
                } catch (const std::exception &e) { errors[g] = e.what(); }
            });
        for (auto &t : workers) t.join();
        for (auto &e : errors) if (!e.empty()) throw std::runtime_error(e);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        m_execMs = *std::max_element(kern.begin(), kern.end());
        m_transferMs = *std::max_element(copy.begin(), copy.end());
        std::cout << "\tdecoded " << n << " frames into pinned memory in " << load_sec << " sec (" << io_threads << " file(s) at a time); device set-up + warm-up " << warm_sec << " sec\n";
        std::cout << "\t" << n << " frames, k=" << k << ", " << G << " device(s): " << sec << " sec, "
                  << (double)n * w * h / 1e6 / sec << " Mpixel/s end to end (host frames in -> host frames out)\n";
        // SaveEXR :1699 / lodepng::encode :1717, straight from the pinned results -- one file per worker thread, like the decode
        const auto te0 = std::chrono::steady_clock::now();
        for_each_file(0, n, [&](int i) {
            const std::string name = "output-animation-" + fs::path(frameNames[i]).stem().string() + (hdr ? ".exr" : ".png");
            if (mid_image_save(out_path(name).c_str(), pin.outs[i], w, h, hdr ? MID_FMT_RGBA32F : MID_FMT_RGBA8)) throw std::runtime_error(mid_last_error());
        });
        if (!hdr) for (int i = 0; i < n; ++i) std::cout << "\t\tencoding png\n";
        std::cout << "\tencoded " << n << " frames in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count()
                  << " sec (" << io_threads << " file(s) at a time)\n";
    }

    void save(std::string name, const Pixel *px, int w, int h, bool hdr) const
    {
        if (hdr) {
            name += ".exr";
            MID_CHECK(mid_image_save(out_path(name).c_str(), px, w, h, MID_FMT_RGBA32F));     // SaveEXR :1699
        } else {
            name += ".png";
            std::cout << "\t\tencoding png\n";
            std::vector<unsigned char> resultData((size_t)w * h * 4);
            const long npx = (long)w * h;
#pragma omp parallel for default(shared) schedule(static)
            for (long i = 0; i < npx; ++i) {                                                          // GetImageFromGPU :97-103
                const float v[4] = {255.0f * px[i].r, 255.0f * px[i].g, 255.0f * px[i].b, 255.0f * px[i].a};
                for (int c = 0; c < 4; ++c)      // truncation; clamped only where the C cast is undefined
                    resultData[i * 4 + c] = !(v[c] > -1.0f) ? 0 : v[c] >= 256.0f ? 255 : (unsigned char)v[c];
            }
            MID_CHECK(mid_image_save(out_path(name).c_str(), resultData.data(), w, h, MID_FMT_RGBA8));  // lodepng::encode :1717
        }
    }

    // RunOnCPU, src/main.cpp:1732-1921
    void RunOnCPU(const std::string &fileName, int numThreads)
    {
        HostImage img = load(fileName, false);
        const int w = img.w, h = img.h;
        std::vector<Pixel> inputPixels((size_t)w * h);
        if (img.format == MID_FMT_RGBA32F) {
            std::cout << "\tloading hdr\n";
            memcpy((void *)inputPixels.data(), img.data(), img.size());
        } else {
            for (size_t i = 0; i < (size_t)w * h; ++i) {                                              // :1804-1807
                inputPixels[i].r = (float)img.data()[4 * i + 0] * (1.0f / 255.0f);
                inputPixels[i].g = (float)img.data()[4 * i + 1] * (1.0f / 255.0f);
                inputPixels[i].b = (float)img.data()[4 * i + 2] * (1.0f / 255.0f);
                inputPixels[i].a = (float)img.data()[4 * i + 3] * (1.0f / 255.0f);
            }
        }
        std::cout << "\tdoing computations\n";
        std::vector<Pixel> outputPixels;
        cpu_bilateral_refpath(inputPixels, w, h, opt.cpu_radius, opt.cpu_sigma_s, opt.cpu_sigma_c, opt.cpu_blue_bug, numThreads, outputPixels);
        std::cout << "\tsaving image\n";
        save("output-cpu", outputPixels.data(), w, h, img.format == MID_FMT_RGBA32F);
    }
};

static void usage()
{
    std::cout <<
        "usage: mi_denoise [image] [options]\n"
        "  image                     target .png or .exr (default Animations/CornellBox/Animation01_LDR_0000.png)\n"
        "  --outdir DIR              where output-*.{png,exr} go (default .)\n"
        "  --device N                HIP device (default 0)\n"
        "  --modes LIST              comma list of bilateral,layers,linear,nlm,multiframe,overlap (default all, reference order)\n"
        "  --gpu-only | --cpu-only   run only the GPU modes / only the CPU runs\n"
        "  --radius R                bilateral window radius (default 20 = TEXEL_WINDOW)\n"
        "  --sigma-s S --sigma-c C   bilateral sigmas (default 2.0 0.2)\n"
        "  --nlm-h H                 NLM filtering parameter (default 0.5)\n"
        "  --search LO,HI --patch LO,HI   half-open NLM ranges (default -7,7 and -3,3; 21x21/7x7 is -10,11 and -3,4)\n"
        "  --temporal-k K            multiframe: frames t-K..t+K of the sorted sequence instead of the reference's list\n"
        "  --animation               denoise EVERY sibling frame with temporal NLM (window +-K, default 2) instead of the mode list\n"
        "  --gpus N                  animation mode: split the sequence into N frame blocks, one per device\n"
        "  --halo host|rccl          animation mode with --gpus N: 'host' (default) streams every block plus its K halo frames from host\n"
        "                            memory through the overlapped pipeline; 'rccl' keeps each block resident in its GPU's HBM and\n"
        "                            exchanges the halo frames GPU to GPU over RCCL/xGMI\n"
        "  --share-device            animation mode: all --gpus N blocks run on --device (a rehearsal of the N-block schedule on fewer\n"
        "                            devices; with --halo rccl it needs a stand-in for RCCL, MID_RCCL_LIBRARY: RCCL itself refuses\n"
        "                            two ranks on one device)\n"
        "  --pageable-host           the six GPU modes: decode into and read back to ordinary (not page-locked) memory; the library\n"
        "                            then moves every copy through its own pinned bounce buffers -- same files, slower copies\n"
        "  --pinned-mb M             animation mode: page-lock at most M MiB of host memory for frames in and out (default 16384);\n"
        "                            frames beyond that, or whose page-locked allocation fails, use pageable memory\n"
        "  --io-threads T            animation mode: decode / encode T files at a time, one per host thread (default min(16, hardware threads))\n"
        "  --cpu-radius R --cpu-sigma-s S --cpu-sigma-c C   CPU path (default 10 10.0 0.2)\n"
        "  --cpu-threads A,B         thread counts of the CPU runs (default 1,8)\n"
        "  --cpu-fix-blue            use the blue channel in the CPU range distance (the reference does not)\n";
}

static bool pair_arg(const char *s, int &a, int &b) { return sscanf(s, "%d,%d", &a, &b) == 2; }

int main(int argc, char **argv)
{
    Options opt;
    bool have_image = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(); exit(EXIT_FAILURE); } return argv[++i]; };
        if (a == "-h" || a == "--help") { usage(); return EXIT_SUCCESS; }
        else if (a == "--outdir") opt.outdir = next();
        else if (a == "--device") opt.device = atoi(next());
        else if (a == "--modes") opt.modes = next();
        else if (a == "--gpu-only") opt.run_cpu = false;
        else if (a == "--cpu-only") opt.run_gpu = false;
        else if (a == "--radius") opt.radius = atoi(next());
        else if (a == "--sigma-s") opt.sigma_s = (float)atof(next());
        else if (a == "--sigma-c") opt.sigma_c = (float)atof(next());
        else if (a == "--nlm-h") opt.nlm_h = (float)atof(next());
        else if (a == "--search") { if (!pair_arg(next(), opt.search_lo, opt.search_hi)) { usage(); return EXIT_FAILURE; } }
        else if (a == "--patch") { if (!pair_arg(next(), opt.patch_lo, opt.patch_hi)) { usage(); return EXIT_FAILURE; } }
        else if (a == "--temporal-k") opt.temporal_k = atoi(next());
        else if (a == "--animation") opt.animation = true;
        else if (a == "--gpus") opt.gpus = atoi(next());
        else if (a == "--share-device") opt.share_device = true;
        else if (a == "--pinned-mb") opt.pinned_mb = atol(next());
        else if (a == "--io-threads") opt.io_threads = atoi(next());
        else if (a == "--pageable-host") opt.pageable_host = true;
        else if (a == "--halo") { const std::string v = next(); if (v == "rccl") opt.halo_rccl = true; else if (v != "host") { usage(); return EXIT_FAILURE; } }
        else if (a == "--cpu-radius") opt.cpu_radius = atoi(next());
        else if (a == "--cpu-sigma-s") opt.cpu_sigma_s = (float)atof(next());
        else if (a == "--cpu-sigma-c") opt.cpu_sigma_c = (float)atof(next());
        else if (a == "--cpu-fix-blue") opt.cpu_blue_bug = false;
        else if (a == "--cpu-threads") {
            opt.cpu_threads.clear();
            std::string s = next();
            size_t pos = 0;
            while (pos < s.size()) { size_t c = s.find(',', pos); if (c == std::string::npos) c = s.size(); opt.cpu_threads.push_back(atoi(s.substr(pos, c - pos).c_str())); pos = c + 1; }
        } else if (!a.empty() && a[0] == '-') { std::cerr << "unknown option " << a << "\n"; usage(); return EXIT_FAILURE; }
        else if (!have_image) { opt.image = a; have_image = true; }
        else { usage(); return EXIT_FAILURE; }
    }
    auto want = [&](const char *m) { return opt.modes == "all" || ("," + opt.modes + ",").find(std::string(",") + m + ",") != std::string::npos; };

    try {
        DenoiseApplication app{opt};
        auto print_time = [&] {                                                   // PRINT_TIME, src/main.cpp:1924-1927
            std::cout << FOREGROUND_COLOR << BACKGROUND_COLOR << "transfer time: " << (unsigned long long)(app.GetTransferMs() * 1e6) << "ns; "
                      << "execution time: " << (unsigned long long)(app.GetExecMs() * 1e6) << "ns\n\n" << CLEAR_COLOR;
        };
        if (opt.animation) {
            std::cout << "######\nRunning on GPU (animation, temporal nonlocal)\n######\n";
            app.RunAnimation();
            print_time();
            return EXIT_SUCCESS;
        }
        if (opt.run_gpu) {
            if (want("bilateral")) { std::cout << "######\nRunning on GPU (nonlinear bialteral)\n######\n"; app.RunOnGPU(false, true, false, false, false); print_time(); }
            if (want("layers")) { std::cout << "######\nRunning on GPU (nonlinear bialteral + layers)\n######\n"; app.RunOnGPU(false, true, false, false, true); print_time(); }
            if (want("linear")) { std::cout << "######\nRunning on GPU (linear bialteral)\n######\n"; app.RunOnGPU(false, false, false, false, false); print_time(); }
            if (want("nlm")) { std::cout << "######\nRunning on GPU (nonlocal)\n######\n"; app.RunOnGPU(true, true, false, false, false); print_time(); }
            if (want("multiframe")) { std::cout << "######\nRunning on GPU (multiframe nonlocal)\n######\n"; app.RunOnGPU(true, true, true, false, false); print_time(); }
            if (want("overlap")) { std::cout << "######\nRunning on GPU (multiframe nonlocal + overlapping)\n######\n"; app.RunOnGPU(true, true, true, true, false); print_time(); }
        }
        if (opt.run_cpu) {
            for (int threads : opt.cpu_threads) {
                const auto t0 = std::chrono::steady_clock::now();
                std::cout << "######\nRunning on CPU (" << threads << (threads == 1 ? " thread" : " threads") << " bialteral)\n######\n";
                app.RunOnCPU(opt.image, threads);
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::cout << FOREGROUND_COLOR << BACKGROUND_COLOR << "Time taken: " << sec << " sec\n\n" << CLEAR_COLOR;   // PRINT_TIME2 :1929-1933
            }
        }
    } catch (const std::exception &e) {                                         // src/main.cpp:1987-1991
        printf("%s\n", e.what());
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
