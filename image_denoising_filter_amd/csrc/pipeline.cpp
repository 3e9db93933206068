// pipeline.cpp -- the frame pipeline: host frames in, denoised host frames out.
//
// Replaces the reference's single-queue ping-pong (RecordCommandsOfOverlappingNLM,
// src/main.cpp:889-989, loop :1539-1573: one command buffer holding "dispatch on texture A"
// followed, without a barrier, by "copy staging -> texture B", descriptor sets swapped by
// parity, and a fence wait after every submit).  Here the stages run concurrently on their own HIP streams:
//
//   upload   : hipMemcpyAsync host frame f -> device ring slot f % RING
//   compute  : temporal NLM of output frame t over ring slots t-k..t+k (two kernel streams, frames alternate)
//   download : hipMemcpyAsync device out slot t % 4 -> host   (RGBA32F outputs, and RGBA8 outputs in pageable memory)
//              RGBA8 outputs in page-locked memory have NO download stage: the kernel's epilogue stores the packed pixels
//              straight into the caller's buffer (4 B per pixel = 17-19 GB/s at the kernel's frame rate, a third of the link)
//
// joined only by events: compute(t) waits for upload(t+k); upload(f) waits for the last
// compute that still reads the slot it overwrites; download(t) waits for compute(t);
// compute(t) waits for download(t-4) before reusing an output slot (no slots and no such wait for direct RGBA8 outputs).  One output frame per launch
// (B = 1 below; coarser batches were measured and lose overlap); the ring holds 2k+4 frames and
// there are four output slots, so uploads run up to three frames ahead of the kernel and
// downloads up to three behind: the stages are decoupled and the slowest one (measured: the
// 33 MB/frame download, 0.70 ms) sets the frame rate.  This is the ONE schedule the library ships
// (a gated chunk-launch alternative was measured in round 2 and removed: LABNOTES.md).  Host frames
// allocated with mid_alloc_host (pinned) are DMA'd directly; pageable memory still works, but it is
// moved through the context's page-locked bounce buffers (csrc/hostcopy.cpp: a host memcpy per 8 MiB
// chunk, never the runtime's pin-on-the-fly path) and the host's memcpy rate -- not the link -- sets
// the pace (uploads on the calling thread, pageable outputs on a helper thread of the call).
//
// Frame selection differs from the reference on purpose: frames are taken
// in the order given, window t-k..t+k clipped at the sequence ends; the reference's "every
// file of the directory in directory order, target twice, last frame never filtered" is not
// reproduced.
#include "common.hpp"
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <thread>
#include <vector>

using namespace mid;

namespace {


// `n` device buffers of at least `bytes` each from the context's cache.  Every pipeline call ends with all four streams
// synchronised, so whatever the cache holds is idle when the next call (serialised by pipe.mu) resizes it.
// Shrink rule: buffers MORE than four times larger than this call needs are given back and reallocated at the size needed (a caller
// that moves from 1080p RGBA32F frames to thumbnails does not keep 400 MB of HBM for them), while alternating between RGBA32F
// and RGBA8 sequences of one frame size -- exactly a factor of four -- keeps the larger set and allocates nothing.
int reserve(mid_pipe_set &s, size_t n, size_t bytes)
{
    if (bytes > s.bytes || (s.bytes > 4 * bytes && !s.p.empty())) {
        for (void *q : s.p) (void)hipFree(q);
        s.p.clear();
        s.bytes = bytes;
    }
    while (s.p.size() < n) { void *q = nullptr; MID_HIP(hipMalloc(&q, s.bytes)); s.p.push_back(q); }
    return MID_OK;
}

int reserve_events(mid_pipe_cache &c, size_t n)
{
    while (c.ev.size() < n) { hipEvent_t e = nullptr; MID_HIP(hipEventCreate(&e)); c.ev.push_back(e); }
    return MID_OK;
}


// A pipeline call that returns early (an error half way) must still leave the context's streams idle: the cached buffers its
// queued work reads and writes belong to the next call the moment this one returns.
struct DrainOnExit {
    mid_ctx *ctx;
    ~DrainOnExit()
    {
        (void)hipStreamSynchronize(ctx->upload);
        (void)hipStreamSynchronize(ctx->compute);
        (void)hipStreamSynchronize(ctx->compute2);
        (void)hipStreamSynchronize(ctx->download);
    }
};

}  // namespace

void mid::pipe_cache_release(mid_ctx *ctx)
{
    mid_pipe_cache &c = ctx->pipe;
    for (mid_pipe_set *s : {&c.ring, &c.out, &c.target, &c.slots, &c.weights, &c.result}) {
        for (void *q : s->p) (void)hipFree(q);
        s->p.clear();
        s->bytes = 0;
    }
    for (hipEvent_t e : c.ev) (void)hipEventDestroy(e);
    c.ev.clear();
    c.last = mid_pipe_last{};
}

// Outputs [first, first+count) of an n-frame host sequence; frames outside that range are only
// uploaded as far as the temporal window needs them (the halo of a frame block).
static int sequence_impl(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                         int n, int k, int first, int count, void *const *host_out, bool out_u8,
                         int overlap, float *timings_ms)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(ctx->compute, "mid_sequence_nlm (four streams, host-side waits)")) return rc;
    MID_REQUIRE(p && host_frames && host_out, "sequence_nlm: NULL argument");
    MID_REQUIRE(n >= 1 && k >= 0 && 2 * k + 2 <= kMaxFrames, "sequence_nlm: bad n=%d k=%d", n, k);
    MID_REQUIRE(first >= 0 && count >= 1 && first + count <= n, "sequence_nlm: bad range first=%d count=%d n=%d", first, count, n);
    MID_REQUIRE(p->width > 0 && p->height > 0, "sequence_nlm: bad size");
    const int f_lo = first - k < 0 ? 0 : first - k;                                  // first frame ever uploaded
    const int f_hi = first + count - 1 + k > n - 1 ? n - 1 : first + count - 1 + k;  // last one
    for (int i = f_lo; i <= f_hi; ++i) MID_REQUIRE(host_frames[i], "sequence_nlm: frame %d is NULL", i);
    for (int i = 0; i < count; ++i) MID_REQUIRE(host_out[i], "sequence_nlm: output %d is NULL", i);

    const size_t npix = (size_t)p->width * p->height;
    const size_t in_bytes = npix * (p->format == MID_FMT_RGBA8 ? 4 : 16);
    const size_t dl_bytes = npix * (out_u8 ? 4 : 16);            // one output frame, as it is written and downloaded
    const int n_up = f_hi - f_lo + 1;
    // Outputs could be filtered in batches of B frames per launch.  Measured on MI355X (16 x 1080p, 21x21/7x7):
    // B=1 2084 Mpixel/s, B=2 1341, B=4 1472, B=8 1403 -- coarser batches bunch the copies and lose overlap
    // while the kernel time barely changes, so one frame per launch it is (the indexing below stays general in B).  Re-measured in
    // round 6 for RGBA8 frames with direct output stores (no download stage to bunch): B = 1 / 2 / 4 4096-4109 / 4106-4121 / 4107-4132
    // Mpixel/s, k = 2 898-904 / 907 / 906 -- the same (profiles/r06_pipe_batch_u8_ab.txt).
    constexpr int B = 1;
    const int nb = (count + B - 1) / B;
    constexpr int DEPTH = 4;                                  // batches in flight per stage
    const int ring = n_up < 2 * k + DEPTH * B ? n_up : 2 * k + DEPTH * B;

    // The clock starts HERE: timings_ms[0] is what the caller waits for, set-up included.  The device ring, the output slots
    // and the events live in the context and are only allocated when a call needs more or larger ones than any call before
    // it (the first call of a context, a larger frame size, a wider window): in a steady stream of sequences nothing is.
    const auto wall0 = std::chrono::steady_clock::now();
    Range call_range("mid_sequence_nlm frames=%d k=%d outputs=[%d,%d)%s", n, k, first, first + count, out_u8 ? " u8" : "");
    std::lock_guard<std::mutex> pipe_lock(ctx->pipe.mu);
    DrainOnExit drain{ctx};
    // Where do the outputs go?  RGBA8 outputs in page-locked memory are written by the kernel itself (`direct`): a pinned buffer is
    // mapped into the device's address space, the packed pixel is 4 B, and at the kernel's frame rate that is 17-19 GB/s of posted
    // writes -- a third of the link -- so the download stage, its output slots and the wait for a free slot disappear.  Measured
    // on 64 x 1080p (profiles/r06_pipeline_u8_timeline.txt, LABNOTES R6.1): staged through hipMemcpyAsync the runtime's copies
    // went from 0.19 to 0.67 ms per frame part way into a call and the four output slots then gated every launch (3230-3590
    // Mpixel/s, 5-10 % spread); direct 4040-4080, spread 1-3 %.  RGBA32F outputs (16 B per pixel: more than the link carries at the
    // kernel's rate) and pageable outputs keep the staged download.
    bool out_pinned = true;
    for (int i = 0; i < count; ++i) out_pinned = out_pinned && host_is_pinned(host_out[i], dl_bytes);
    bool direct = out_u8 && out_pinned;
    for (int i = 0; direct && i < count; ++i) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, host_out[i], 0) != hipSuccess || !dp) { (void)hipGetLastError(); direct = false; }
    }
    mid_pipe_set &dring = ctx->pipe.ring, &dout = ctx->pipe.out;
    if (int rc = reserve(dring, ring, in_bytes)) return rc;
    if (!direct) { if (int rc = reserve(dout, DEPTH * B, dl_bytes)) return rc; }
    if (int rc = reserve_events(ctx->pipe, 2 * (size_t)n_up + 4 * (size_t)nb)) return rc;
    ctx->pipe.last = mid_pipe_last{};
    struct Events { hipEvent_t *ev; } up0{ctx->pipe.ev.data()}, up1{up0.ev + n_up}, c0{up1.ev + n_up}, c1{c0.ev + nb}, d0{c1.ev + nb}, d1{d0.ev + nb};
    auto slot = [&](int f) { return dring.p[(f - f_lo) % ring]; };

    int next_upload = f_lo;
    auto upload = [&](int f) -> int {
        Range r("upload %d", f);
        if (f - f_lo >= ring) {   // the slot still holds frame f-ring, read by outputs (f-ring)-k .. (f-ring)+k
            int last_reader = f - ring + k;
            if (last_reader > first + count - 1) last_reader = first + count - 1;
            if (last_reader >= first) {
                // Batches alternate between the two kernel streams, so the readers of this slot sit on BOTH:
                // batch lb and every earlier same-parity batch are ordered before c1[lb] by stream order, the
                // other-parity readers (lb-1, lb-3, ...) before c1[lb-1].  Waiting on both makes the overwrite
                // safe by construction, not by the kernels happening to finish in launch order.
                const int lb = (last_reader - first) / B;
                MID_HIP(hipStreamWaitEvent(ctx->upload, c1.ev[lb], 0));
                if (lb >= 1) MID_HIP(hipStreamWaitEvent(ctx->upload, c1.ev[lb - 1], 0));
            }
        }
        MID_HIP(hipEventRecord(up0.ev[f - f_lo], ctx->upload));
        if (int rc = copy_h2d(ctx, slot(f), host_frames[f], in_bytes, ctx->upload)) return rc;
        MID_HIP(hipEventRecord(up1.ev[f - f_lo], ctx->upload));
        return MID_OK;
    };
    // Pinned outputs: download(bi) is queued right behind compute(bi) and the host runs on.  Pageable outputs: copy_d2h returns
    // only when the frame is in the caller's buffer (a host memcpy per chunk out of the bounce buffers), so those downloads run on
    // a helper thread of this call: the calling thread keeps bouncing uploads in and queueing launches while the helper copies
    // finished frames out -- the two memcpy streams overlap (16 x 1080p RGBA32F with pageable frames on both sides: 42.9 ms with
    // both on one thread).  The helper takes batch bi once compute(bi) has been queued; compute(bi) is queued only when the helper
    // has recorded d1[bi-DEPTH] (the output slot is free) -- the event order of the pinned case, kept by two counters.
    auto download = [&](int bi) -> int {
        const int b0 = first + bi * B, bn = (first + count - b0) < B ? (first + count - b0) : B;
        Range r("download %d", b0);
        MID_HIP(hipStreamWaitEvent(ctx->download, c1.ev[bi], 0));
        MID_HIP(hipEventRecord(d0.ev[bi], ctx->download));
        for (int i = 0; i < bn; ++i)
            if (int rc = copy_d2h(ctx, host_out[b0 - first + i], dout.p[(bi % DEPTH) * B + i], dl_bytes, ctx->download)) return rc;
        MID_HIP(hipEventRecord(d1.ev[bi], ctx->download));
        return MID_OK;
    };
    struct Helper {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        int issued = -1;                 // highest batch whose launch has been queued          (caller -> helper)
        int done = -1;                   // highest batch whose download has finished, d1 recorded (helper -> caller)
        bool stop = false;
        int rc = MID_OK;
        char err[512] = "";
        ~Helper()
        {
            if (!th.joinable()) return;
            { std::lock_guard<std::mutex> l(mu); stop = true; }
            cv.notify_all();
            th.join();                   // before DrainOnExit and before anything the download lambda refers to goes away
        }
    } helper;
    bool threaded = !out_pinned && overlap;
    if (threaded) try {
        helper.th = std::thread([&] {
            (void)hipSetDevice(ctx->device);
            for (int bi = 0; bi < nb; ++bi) {
                {
                    std::unique_lock<std::mutex> l(helper.mu);
                    helper.cv.wait(l, [&] { return helper.stop || helper.issued >= bi; });
                    if (helper.issued < bi) return;          // stopped before this batch was launched
                }
                const int rc = download(bi);
                {
                    std::lock_guard<std::mutex> l(helper.mu);
                    if (rc) { helper.rc = rc; snprintf(helper.err, sizeof helper.err, "%s", mid_last_error()); helper.stop = true; }
                    else helper.done = bi;
                }
                helper.cv.notify_all();
                if (rc) return;
            }
        });
    } catch (...) {
        threaded = false;            // no thread to be had (EAGAIN under a thread limit): the downloads run on the calling thread
    }
    // the helper's error, if any, becomes this thread's error
    auto helper_failed = [&]() -> int {
        std::lock_guard<std::mutex> l(helper.mu);
        return helper.rc ? set_error(helper.rc, "%s", helper.err) : MID_OK;
    };

    for (int bi = 0; bi < nb; ++bi) {
        const int b0 = first + bi * B, bn = (first + count - b0) < B ? (first + count - b0) : B;
        const int need = b0 + bn - 1 + k < n - 1 ? b0 + bn - 1 + k : n - 1;
        const int ahead = overlap ? (need + (DEPTH - 1) * B < f_hi ? need + (DEPTH - 1) * B : f_hi) : need;
        // frames up to the batch's last window must be resident; with overlap the next batches' frames are
        // started now as well: they only wait for kernels already enqueued and ride beside this one
        while (next_upload <= need) { if (int rc = upload(next_upload++)) return rc; }
        // Consecutive batches go to alternating kernel streams: every dependency between them is an explicit
        // event (inputs, ring-slot reuse, output-slot reuse), so the tail of one launch -- 1156 workgroups on 512
        // slots leave the last round a quarter full -- overlaps the head of the next instead of idling the CUs.
        // (measured: one kernel stream 2385, two 2575, three 1892, four 1652 Mpixel/s; giving the second stream a
        // lower or higher priority than the first: 2410-2430; splitting every download over two copy streams: 1860)
        hipStream_t cs = overlap && (bi & 1) ? ctx->compute2 : ctx->compute;
        MID_HIP(hipStreamWaitEvent(cs, up1.ev[need - f_lo], 0));
        // Output slot bi % DEPTH was written by batch bi-DEPTH and is read only by download(bi-DEPTH), which
        // itself waited for c1[bi-DEPTH]: d1[bi-DEPTH] therefore orders both the writer and the only reader of
        // the slot before this batch, whichever kernel stream they ran on.
        if (bi >= DEPTH && !direct) {
            if (threaded) {              // d1[bi-DEPTH] must have been RECORDED before a stream can be told to wait for it
                std::unique_lock<std::mutex> l(helper.mu);
                helper.cv.wait(l, [&] { return helper.stop || helper.done >= bi - DEPTH; });
                if (helper.rc) { l.unlock(); return helper_failed(); }
            }
            MID_HIP(hipStreamWaitEvent(cs, d1.ev[bi - DEPTH], 0));
        }

        const int lo = b0 - k < 0 ? 0 : b0 - k;
        const void *tbl[kMaxFrames];
        for (int f = lo; f <= need; ++f) tbl[f - lo] = slot(f);
        mid_pixel *o[kMaxFrames];
        for (int i = 0; i < bn; ++i) {
            if (direct) MID_HIP(hipHostGetDevicePointer((void **)&o[i], host_out[b0 - first + i], 0));
            else o[i] = (mid_pixel *)dout.p[(bi % DEPTH) * B + i];
        }
        {
            Range nlm_range("nlm %d", b0);
            MID_HIP(hipEventRecord(c0.ev[bi], cs));
            // out_u8: GetImageFromGPU's u8 conversion (src/main.cpp:97-103) in the kernel's epilogue -- a quarter of the
            // bytes to write and to download
            if (int rc = nlm_temporal_out(ctx, p, tbl, need - lo + 1, k, b0 - lo, bn, (void *const *)o, out_u8 ? 1 : 0, cs, 1)) return rc;
            MID_HIP(hipEventRecord(c1.ev[bi], cs));
        }

        while (next_upload <= ahead) { if (int rc = upload(next_upload++)) return rc; }

        if (direct) {
            // nothing to download: the launch wrote the caller's buffer
        } else if (threaded) {
            { std::lock_guard<std::mutex> l(helper.mu); helper.issued = bi; }
            helper.cv.notify_all();
        } else if (int rc = download(bi)) return rc;

        if (!overlap) {   // the reference's behaviour: a fence wait after every submit (src/main.cpp:1092)
            MID_HIP(hipStreamSynchronize(ctx->upload));
            MID_HIP(hipStreamSynchronize(ctx->compute));
            MID_HIP(hipStreamSynchronize(ctx->download));
        }
    }
    if (threaded) {
        {
            std::unique_lock<std::mutex> l(helper.mu);
            helper.cv.wait(l, [&] { return helper.stop || helper.done >= nb - 1; });
        }
        helper.th.join();
        if (int rc = helper_failed()) return rc;
    }
    {
        Range r("drain");
        MID_HIP(hipStreamSynchronize(ctx->upload));
        MID_HIP(hipStreamSynchronize(ctx->compute));
        MID_HIP(hipStreamSynchronize(ctx->compute2));
        MID_HIP(hipStreamSynchronize(ctx->download));
    }
    const auto wall1 = std::chrono::steady_clock::now();
    ctx->pipe.last = {n_up, nb, f_lo, first, B, direct};

    if (timings_ms) {
        float kern = 0.f, copy = 0.f, ms = 0.f;
        for (int bi = 0; bi < nb; ++bi) {
            MID_HIP(hipEventElapsedTime(&ms, c0.ev[bi], c1.ev[bi])); kern += ms;
            if (!direct) { MID_HIP(hipEventElapsedTime(&ms, d0.ev[bi], d1.ev[bi])); copy += ms; }
        }
        for (int i = 0; i < n_up; ++i) { MID_HIP(hipEventElapsedTime(&ms, up0.ev[i], up1.ev[i])); copy += ms; }
        timings_ms[0] = std::chrono::duration<float, std::milli>(wall1 - wall0).count();
        timings_ms[1] = kern;
        timings_ms[2] = copy;
    }
    return MID_OK;
}

// Device timeline of the context's last mid_sequence_nlm* call, read back from the events the call left in the context's
// cache (valid until the next pipeline call on this context; no profiler involved, so the call ran at its own pace).
extern "C" int mid_pipe_last_timeline(mid_ctx *ctx, int cap, float *upload_ms, int *n_uploads, int *first_upload_frame,
                                      float *output_ms, int *n_outputs, int *first_output_frame)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(upload_ms && n_uploads && first_upload_frame && output_ms && n_outputs && first_output_frame && cap >= 0,
                "pipe_last_timeline: bad argument");
    std::lock_guard<std::mutex> pipe_lock(ctx->pipe.mu);
    const mid_pipe_last &L = ctx->pipe.last;
    MID_REQUIRE(L.n_up > 0, "pipe_last_timeline: no mid_sequence_nlm* call has completed on this context");
    MID_REQUIRE(cap >= L.n_up && cap >= L.nb, "pipe_last_timeline: cap=%d, need %d uploads and %d outputs", cap, L.n_up, L.nb);
    MID_REQUIRE(ctx->pipe.ev.size() >= 2 * (size_t)L.n_up + 4 * (size_t)L.nb, "pipe_last_timeline: the event cache was released");
    hipEvent_t *up0 = ctx->pipe.ev.data(), *up1 = up0 + L.n_up, *c0 = up1 + L.n_up, *c1 = c0 + L.nb, *d0 = c1 + L.nb, *d1 = d0 + L.nb;
    for (int i = 0; i < L.n_up; ++i) {
        MID_HIP(hipEventElapsedTime(&upload_ms[2 * i], up0[0], up0[i]));
        MID_HIP(hipEventElapsedTime(&upload_ms[2 * i + 1], up0[0], up1[i]));
    }
    for (int i = 0; i < L.nb; ++i) {
        MID_HIP(hipEventElapsedTime(&output_ms[4 * i], up0[0], c0[i]));
        MID_HIP(hipEventElapsedTime(&output_ms[4 * i + 1], up0[0], c1[i]));
        // direct RGBA8 outputs have no download stage: reported as an empty interval at the end of the launch
        MID_HIP(hipEventElapsedTime(&output_ms[4 * i + 2], up0[0], L.direct ? c1[i] : d0[i]));
        MID_HIP(hipEventElapsedTime(&output_ms[4 * i + 3], up0[0], L.direct ? c1[i] : d1[i]));
    }
    *n_uploads = L.n_up; *first_upload_frame = L.f_lo; *n_outputs = L.nb; *first_output_frame = L.first;
    return MID_OK;
}

extern "C" int mid_sequence_nlm_range(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                                      int n, int k, int first, int count, mid_pixel *const *host_out,
                                      int overlap, float *timings_ms)
{
    return sequence_impl(ctx, p, host_frames, n, k, first, count, (void *const *)host_out, false, overlap, timings_ms);
}

extern "C" int mid_sequence_nlm_range_u8(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                                         int n, int k, int first, int count, uint8_t *const *host_out,
                                         int overlap, float *timings_ms)
{
    return sequence_impl(ctx, p, host_frames, n, k, first, count, (void *const *)host_out, true, overlap, timings_ms);
}

extern "C" int mid_sequence_nlm(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                                int n, int k, mid_pixel *const *host_out, int overlap, float *timings_ms)
{
    return mid_sequence_nlm_range(ctx, p, host_frames, n, k, 0, n, host_out, overlap, timings_ms);
}

// The reference's own multi-frame mode: target fixed, neighbours streamed (see mi_denoise.h).
extern "C" int mid_nlm_multiframe(mid_ctx *ctx, const mid_nlm_params *p, const void *host_target,
                                  const void *const *host_frames, int n, mid_pixel *host_out,
                                  int overlap, float *timings_ms)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(ctx->compute, "mid_nlm_multiframe (three streams, host-side waits)")) return rc;
    MID_REQUIRE(p && host_target && host_frames && host_out, "nlm_multiframe: NULL argument");
    MID_REQUIRE(n >= 1 && p->width > 0 && p->height > 0, "nlm_multiframe: bad n=%d or size", n);
    for (int i = 0; i < n; ++i) MID_REQUIRE(host_frames[i], "nlm_multiframe: frame %d is NULL", i);
    const size_t npix = (size_t)p->width * p->height;
    const size_t in_bytes = npix * (p->format == MID_FMT_RGBA8 ? 4 : 16), out_bytes = npix * 16;

    // (clock and cached buffers as in sequence_impl: timings_ms[0] covers the whole call)
    const auto wall0 = std::chrono::steady_clock::now();
    Range call_range("mid_nlm_multiframe frames=%d overlap=%d", n, overlap);
    std::lock_guard<std::mutex> pipe_lock(ctx->pipe.mu);
    DrainOnExit drain{ctx};
    mid_pipe_set &dtarget = ctx->pipe.target, &dslot = ctx->pipe.slots, &dW = ctx->pipe.weights, &dout = ctx->pipe.result;
    if (int rc = reserve(dtarget, 1, in_bytes)) return rc;
    constexpr int SLOTS = 3;                                   // neighbour frames in flight: uploads run two dispatches ahead
    if (int rc = reserve(dslot, n < SLOTS ? n : SLOTS, in_bytes)) return rc;
    if (int rc = reserve(dW, 1, npix * sizeof(mid_weightinfo))) return rc;
    if (int rc = reserve(dout, 1, out_bytes)) return rc;
    if (int rc = reserve_events(ctx->pipe, 4 * (size_t)n + 4)) return rc;
    ctx->pipe.last = mid_pipe_last{};                         // the cached events are about to be re-recorded in another layout
    struct Events { hipEvent_t *ev; } up0{ctx->pipe.ev.data()}, up1{up0.ev + n}, c0{up1.ev + n}, c1{c0.ev + n}, misc{c1.ev + n};

    // target + cleared weight buffer (the reference relies on a fresh allocation being zero)
    MID_HIP(hipEventRecord(misc.ev[0], ctx->upload));
    if (int rc = copy_h2d(ctx, dtarget.p[0], host_target, in_bytes, ctx->upload)) return rc;
    MID_HIP(hipEventRecord(misc.ev[1], ctx->upload));
    MID_HIP(hipMemsetAsync(dW.p[0], 0, npix * sizeof(mid_weightinfo), ctx->compute));
    MID_HIP(hipStreamWaitEvent(ctx->compute, misc.ev[1], 0));

    auto upload = [&](int f) -> int {
        Range r("upload %d", f);
        if (f >= SLOTS) MID_HIP(hipStreamWaitEvent(ctx->upload, c1.ev[f - SLOTS], 0));   // slot still read by dispatch f-SLOTS
        MID_HIP(hipEventRecord(up0.ev[f], ctx->upload));
        if (int rc = copy_h2d(ctx, dslot.p[f % SLOTS], host_frames[f], in_bytes, ctx->upload)) return rc;
        MID_HIP(hipEventRecord(up1.ev[f], ctx->upload));
        return MID_OK;
    };
    int next_upload = 0;
    for (int f = 0; f < n; ++f) {
        while (next_upload <= f) { if (int rc = upload(next_upload++)) return rc; }
        {
            Range r("nlm %d", f);
            MID_HIP(hipStreamWaitEvent(ctx->compute, up1.ev[f], 0));
            MID_HIP(hipEventRecord(c0.ev[f], ctx->compute));
            if (int rc = mid_nlm_accum(ctx, p, dtarget.p[0], dslot.p[f % SLOTS], (mid_weightinfo *)dW.p[0], ctx->compute)) return rc;
            MID_HIP(hipEventRecord(c1.ev[f], ctx->compute));
        }
        // overlap: frames f+1, f+2 ride beside dispatch f (their slots were last read by dispatches already enqueued).  They are
        // started AFTER dispatch f has been queued, as in sequence_impl: a pageable frame blocks the calling thread inside
        // copy_h2d until its chunks have gone through the bounce buffers, and the device must not sit idle meanwhile.
        const int ahead = overlap ? (f + SLOTS - 1 < n - 1 ? f + SLOTS - 1 : n - 1) : f;
        while (next_upload <= ahead) { if (int rc = upload(next_upload++)) return rc; }
        if (!overlap) {   // fence after every submit, src/main.cpp:1092
            MID_HIP(hipStreamSynchronize(ctx->compute));
            MID_HIP(hipStreamSynchronize(ctx->upload));
        }
    }
    mid_normalize_params np{p->width, p->height};
    if (int rc = mid_normalize(ctx, &np, (const mid_weightinfo *)dW.p[0], (mid_pixel *)dout.p[0], ctx->compute)) return rc;
    MID_HIP(hipEventRecord(misc.ev[2], ctx->compute));
    if (int rc = copy_d2h(ctx, host_out, dout.p[0], out_bytes, ctx->compute)) return rc;
    MID_HIP(hipEventRecord(misc.ev[3], ctx->compute));
    MID_HIP(hipStreamSynchronize(ctx->upload));
    MID_HIP(hipStreamSynchronize(ctx->compute));
    const auto wall1 = std::chrono::steady_clock::now();
    if (timings_ms) {
        float kern = 0.f, copy = 0.f, ms = 0.f;
        for (int f = 0; f < n; ++f) {
            MID_HIP(hipEventElapsedTime(&ms, c0.ev[f], c1.ev[f])); kern += ms;
            MID_HIP(hipEventElapsedTime(&ms, up0.ev[f], up1.ev[f])); copy += ms;
        }
        MID_HIP(hipEventElapsedTime(&ms, misc.ev[0], misc.ev[1])); copy += ms;
        MID_HIP(hipEventElapsedTime(&ms, misc.ev[2], misc.ev[3])); copy += ms;
        MID_HIP(hipEventElapsedTime(&ms, c1.ev[n - 1], misc.ev[2])); kern += ms;   // normalize
        timings_ms[0] = std::chrono::duration<float, std::milli>(wall1 - wall0).count();
        timings_ms[1] = kern;
        timings_ms[2] = copy;
    }
    return MID_OK;
}
