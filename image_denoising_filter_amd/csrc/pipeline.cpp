// pipeline.cpp -- the frame pipeline: host frames in, denoised host frames out.
//
// Replaces the reference's single-queue ping-pong (RecordCommandsOfOverlappingNLM,
// src/main.cpp:889-989, loop :1539-1573: one command buffer holding "dispatch on texture A"
// followed, without a barrier, by "copy staging -> texture B", descriptor sets swapped by
// parity, and a fence wait after every submit).  Here the stages run concurrently on their own HIP streams:
//
//   upload   : hipMemcpyAsync host frame f -> device ring slot f % RING
//   compute  : temporal NLM of output frame t over ring slots t-k..t+k (two kernel streams, frames alternate)
//   download : hipMemcpyAsync device out slot t % 2 -> host
//
// joined only by events: compute(t) waits for upload(t+k); upload(f) waits for the last
// compute that still reads the slot it overwrites; download(t) waits for compute(t);
// compute(t) waits for download(t-4) before reusing an output slot.  Outputs go in batches of
// B frames per launch; the ring holds 2k+4B frames and there are four output slots, so uploads
// run up to three batches ahead of the kernel and downloads up to three behind: the stages are
// decoupled and the slowest one (measured: the 33 MB/frame download, 0.70 ms) sets the frame
// rate.  Host frames
// allocated with mid_alloc_host (pinned) are DMA'd directly; pageable memory still works but
// HIP stages it and the overlap is lost.
//
// Frame selection differs from the reference on purpose: frames are taken
// in the order given, window t-k..t+k clipped at the sequence ends; the reference's "every
// file of the directory in directory order, target twice, last frame never filtered" is not
// reproduced.
#include "common.hpp"
#include <chrono>
#include <cstdlib>
#include <vector>

using namespace mid;

namespace {

struct EventPool {
    std::vector<hipEvent_t> ev;
    ~EventPool() { for (auto e : ev) (void)hipEventDestroy(e); }
    int make(size_t n)
    {
        ev.resize(n, nullptr);
        for (auto &e : ev) MID_HIP(hipEventCreate(&e));
        return MID_OK;
    }
};

struct DeviceBufs {
    std::vector<void *> p;
    ~DeviceBufs() { for (auto q : p) (void)hipFree(q); }
    int make(size_t n, size_t bytes)
    {
        p.assign(n, nullptr);
        for (auto &q : p) MID_HIP(hipMalloc(&q, bytes));
        return MID_OK;
    }
};

}  // namespace

// ---- gated pipeline (opt-in: MID_PIPE_GATED=1) ------------------------------------------------------------------
// A second schedule for the same job, built to test whether the per-frame launches of the event-joined pipeline below
// cost throughput (a 1080p frame is 2.26 rounds of workgroups: one launch per frame takes 0.75 ms against 0.585 ms per
// frame inside a 16-frame launch).  Here the launches are chunk-sized (up to kGateChunk output frames) and the
// dependency on the uploads moves from the launch into the kernel: the launch is enqueued straight away, a workgroup
// of output t waits on a device word the upload stream raises behind frame t+k's copy (csrc/nlm.hip, gate_wait: bounded
// spin), and the workgroup that finishes the last tile of an output frame raises a word in pinned host memory; this
// thread sees it and queues that frame's download.  Frames flow upload -> filter -> download one by one although the
// launches are chunk-sized, and the CUs never drain between frames.
// MEASURED (tools/pipe_gated_ab.py, 1080p, 21x21/7x7, k=0): no gain.  16 frames: RGBA8 2812 vs 2821 Mpixel/s event-joined,
// RGBA32F 2609 vs 2565; 64 frames: 3261 vs 3249 and 2854 vs 2847.  The two co-running per-frame launches of the
// event-joined pipeline already keep the CUs full (at 64 frames it runs at 0.638 ms per frame = the 0.584 ms of a warm
// 64-frame launch plus the clock ramp below), and what separates the 16-frame figures from the kernel's batched rate is
// the GPU's clock ramp after idle: ONE 64-frame launch takes 36.8 ms back to back and 39.6 ms after 50 ms of idle.
// The event-joined pipeline therefore stays the default (no spinning workgroups, no mid-kernel visibility rules);
// this one is kept, tested bit for bit against it, as the measured alternative.
//
// Invariants the kernel relies on (kept here by construction):
//   * uploads are queued in frame order on ONE stream, each followed by its flag write, so "flag of frame f is up"
//     implies every frame <= f has landed;
//   * a device slot is written at most once while a launch that reads it is in flight, and only before that launch
//     reads it: frame g reuses the slot of frame g-RS, RS = 2*chunk + 2k, and waits for the launch that last read
//     g-RS -- the running launch (the next chunk) only reads frames < g;
//   * an output slot is rewritten two chunks later, after the host has queued (and the kernel's stream has waited
//     for) the download of its previous content.
// Every wait is bounded: the kernel's spins time out after 4 s and raise the abort word, and the host's polls stop as
// soon as the launch they wait for has completed without delivering (which can only mean an abort).
namespace {
constexpr int kGateChunk = 32;
constexpr int kUploadAhead = 3;

struct HostBuf {   // pinned host memory, released on every path
    void *p = nullptr;
    ~HostBuf() { if (p) (void)hipHostFree(p); }
    int make(size_t bytes) { MID_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault)); memset(p, 0, bytes); return MID_OK; }
};
}  // namespace

static int sequence_gated(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames, int n, int k,
                          int first, int count, void *const *host_out, bool out_u8, float *timings_ms)
{
    const int f_lo = first - k < 0 ? 0 : first - k;
    const int f_hi = first + count - 1 + k > n - 1 ? n - 1 : first + count - 1 + k;
    const size_t npix = (size_t)p->width * p->height;
    const size_t in_bytes = npix * (p->format == MID_FMT_RGBA8 ? 4 : 16);
    const size_t dl_bytes = npix * (out_u8 ? 4 : 16);
    const int n_up = f_hi - f_lo + 1;
    int C = count < kGateChunk ? count : kGateChunk;
    if (const char *e = getenv("MID_PIPE_CHUNK")) { const int v = atoi(e); if (v >= 1 && v < C) C = v; }   // development: smaller chunks
    if (C + 2 * k > kMaxFrames) C = kMaxFrames - 2 * k;
    const int nch = (count + C - 1) / C;
    const int RS = n_up < 2 * C + 2 * k ? n_up : 2 * C + 2 * k;       // input slots
    const int OS = count < 2 * C ? count : 2 * C;                     // output slots

    DeviceBufs dring, dout, dflags;
    if (int rc = dring.make(RS, in_bytes)) return rc;
    if (int rc = dout.make(OS, dl_bytes)) return rc;
    // device words: ready[n_up] | abort | pad to a 128-byte line | done[count][kGateDoneWordsPerFrame]
    const size_t done_off = ((size_t)n_up + 1 + 31) / 32 * 32;
    const size_t n_words = done_off + (size_t)count * kGateDoneWordsPerFrame;
    if (int rc = dflags.make(1, n_words * sizeof(uint32_t))) return rc;
    uint32_t *ready = (uint32_t *)dflags.p[0], *abort_w = ready + n_up, *done = ready + done_off;
    HostBuf hflags;                                                   // host_done[count] | one | abort read-back
    if (int rc = hflags.make(((size_t)count + 2) * sizeof(uint32_t))) return rc;
    volatile uint32_t *host_done = (volatile uint32_t *)hflags.p;
    uint32_t *one = (uint32_t *)hflags.p + count, *abort_rb = one + 1;
    *one = 1u;

    EventPool up0, up1, k0, k1, d0, d1, dld;
    for (EventPool *e : {&up0, &up1}) if (int rc = e->make(n_up)) return rc;
    for (EventPool *e : {&k0, &k1, &dld}) if (int rc = e->make(nch)) return rc;
    for (EventPool *e : {&d0, &d1}) if (int rc = e->make(count)) return rc;
    auto slot = [&](int f) { return dring.p[(f - f_lo) % RS]; };
    auto chunk_of_output = [&](int t) { return (t - first) / C; };

    const auto wall0 = std::chrono::steady_clock::now();
    MID_HIP(hipMemsetAsync(ready, 0, n_words * sizeof(uint32_t), ctx->upload));
    hipEvent_t zeroed = nullptr;
    MID_HIP(hipEventCreate(&zeroed));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } zguard{zeroed};
    MID_HIP(hipEventRecord(zeroed, ctx->upload));
    MID_HIP(hipStreamWaitEvent(ctx->compute, zeroed, 0));           // counters and abort word are zero before any launch

    // One host loop drives everything that cannot be queued up front: it launches a chunk, then alternates between
    // queueing the next upload (one at a time: a copy from pageable memory blocks the caller, and downloads must not
    // wait behind a whole chunk of those) and looking for finished outputs to download.
    int next_upload = f_lo, upload_limit = f_lo - 1;                  // frames <= upload_limit are needed by launched chunks
    auto upload_one = [&]() -> int {
        const int f = next_upload++;
        if (f - f_lo >= RS) {                                       // slot still holds frame f-RS: wait for its last reader's launch
            int last_reader = f - RS + k;
            if (last_reader > first + count - 1) last_reader = first + count - 1;
            if (last_reader >= first) MID_HIP(hipStreamWaitEvent(ctx->upload, k1.ev[chunk_of_output(last_reader)], 0));
        }
        MID_HIP(hipEventRecord(up0.ev[f - f_lo], ctx->upload));
        MID_HIP(hipMemcpyAsync(slot(f), host_frames[f], in_bytes, hipMemcpyHostToDevice, ctx->upload));
        MID_HIP(hipEventRecord(up1.ev[f - f_lo], ctx->upload));
        // the gate: queued behind the frame's own copy on the same stream
        if (int rc = gate_raise(ready + (f - f_lo), ctx->upload)) return rc;
        return MID_OK;
    };
    auto launch_chunk = [&](int j) -> int {                           // the launch goes first: it needs no upload to be queued, only the slots' addresses
        const int c0 = first + j * C, cn = (first + count - c0) < C ? (first + count - c0) : C;
        const int lo = c0 - k < f_lo ? f_lo : c0 - k;
        const int need = c0 + cn - 1 + k > f_hi ? f_hi : c0 + cn - 1 + k;
        if (j >= 2) MID_HIP(hipStreamWaitEvent(ctx->compute, dld.ev[j - 2], 0));   // output slots of chunk j-2 have been downloaded
        const void *tbl[kMaxFrames];
        for (int f = lo; f <= need; ++f) tbl[f - lo] = slot(f);
        void *o[kMaxFrames];
        for (int i = 0; i < cn; ++i) o[i] = dout.p[(c0 - first + i) % OS];
        GateArgs g{ready + (lo - f_lo), done + (size_t)(c0 - first) * kGateDoneWordsPerFrame, (uint32_t *)host_done + (c0 - first), abort_w};
        MID_HIP(hipEventRecord(k0.ev[j], ctx->compute));
        if (int rc = nlm_temporal_out(ctx, p, tbl, need - lo + 1, k, c0 - lo, cn, o, out_u8 ? 1 : 0, ctx->compute, &g)) return rc;
        MID_HIP(hipEventRecord(k1.ev[j], ctx->compute));
        upload_limit = need;
        return MID_OK;
    };

    int rc = launch_chunk(0);
    int launched = 1;
    if (!rc && nch > 1) { rc = launch_chunk(1); launched = 2; }
    bool stalled = false;
    int next_dl = 0;                                                  // next output (relative to `first`) to download
    unsigned idle = 0;
    while (!rc && !stalled && next_dl < count) {
        bool progress = false;
        // uploads run at most kUploadAhead frames ahead of the output being waited for: copies are served in submission
        // order, and a download queued behind a whole chunk of uploads waited for all of them (measured: the first
        // 33 MB download took 9 ms behind 16 queued uploads)
        if (next_upload <= upload_limit && next_upload <= first + next_dl + k + kUploadAhead) { rc = upload_one(); progress = true; if (rc) break; }
        if (__atomic_load_n((const uint32_t *)&host_done[next_dl], __ATOMIC_ACQUIRE) != 0u) {
            const int i = next_dl++;
            const int j = i / C;
            rc = [&]() -> int {
                MID_HIP(hipEventRecord(d0.ev[i], ctx->download));
                MID_HIP(hipMemcpyAsync(host_out[i], dout.p[i % OS], dl_bytes, hipMemcpyDeviceToHost, ctx->download));
                MID_HIP(hipEventRecord(d1.ev[i], ctx->download));
                if (next_dl == count || next_dl / C != j) {           // chunk j is out: its output slots may be rewritten, the next chunk may be launched
                    MID_HIP(hipEventRecord(dld.ev[j], ctx->download));
                    if (launched < nch) { if (int r = launch_chunk(launched)) return r; ++launched; }
                }
                return MID_OK;
            }();
            progress = true;
        }
        if (progress) { idle = 0; continue; }
        if ((++idle & 0x3ffu) == 0u) {
            // nothing to queue, nothing finished: fine while the launch that owns output next_dl is still running; once it
            // has completed without delivering (only an abort or a fault does that) there is nothing left to wait for
            const hipError_t q = hipEventQuery(k1.ev[next_dl / C]);
            if (q != hipErrorNotReady && __atomic_load_n((const uint32_t *)&host_done[next_dl], __ATOMIC_ACQUIRE) == 0u) stalled = true;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (stalled || rc) {   // make every workgroup that is still waiting leave, then drain
        (void)hipMemcpyAsync(abort_w, one, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->download);
    }
    const hipError_t s1 = hipStreamSynchronize(ctx->upload), s2 = hipStreamSynchronize(ctx->compute), s3 = hipStreamSynchronize(ctx->download);
    if (rc) return rc;
    MID_HIP(s1); MID_HIP(s2); MID_HIP(s3);
    MID_HIP(hipMemcpy(abort_rb, abort_w, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (stalled || *abort_rb != 0u)
        return set_error(MID_ERR_HIP, "sequence_nlm: the gated pipeline stalled (a frame's upload never became visible to the kernel within %.0f s); no output is valid", 4.0);
    const auto wall1 = std::chrono::steady_clock::now();

    if (getenv("MID_PIPE_TRACE")) {
        for (int j = 0; j < nch; ++j) {
            float a0, a1;
            (void)hipEventElapsedTime(&a0, up0.ev[0], k0.ev[j]); (void)hipEventElapsedTime(&a1, up0.ev[0], k1.ev[j]);
            fprintf(stderr, "gated launch %d  %.3f-%.3f ms\n", j, a0, a1);
        }
        for (int i = 0; i < count; ++i) {
            float u1 = 0.f, e0, e1;
            const int fu = first + i + k > f_hi ? f_hi : first + i + k;
            (void)hipEventElapsedTime(&u1, up0.ev[0], up1.ev[fu - f_lo]);
            (void)hipEventElapsedTime(&e0, up0.ev[0], d0.ev[i]); (void)hipEventElapsedTime(&e1, up0.ev[0], d1.ev[i]);
            fprintf(stderr, "output %2d  last input up at %.3f  down %.3f-%.3f ms\n", i, u1, e0, e1);
        }
    }
    if (timings_ms) {
        float kern = 0.f, copy = 0.f, ms = 0.f;
        for (int j = 0; j < nch; ++j) { MID_HIP(hipEventElapsedTime(&ms, k0.ev[j], k1.ev[j])); kern += ms; }
        for (int i = 0; i < count; ++i) { MID_HIP(hipEventElapsedTime(&ms, d0.ev[i], d1.ev[i])); copy += ms; }
        for (int i = 0; i < n_up; ++i) { MID_HIP(hipEventElapsedTime(&ms, up0.ev[i], up1.ev[i])); copy += ms; }
        timings_ms[0] = std::chrono::duration<float, std::milli>(wall1 - wall0).count();
        timings_ms[1] = kern;      // launches are chunk-sized and include their waits for the uploads
        timings_ms[2] = copy;
    }
    return MID_OK;
}

// Outputs [first, first+count) of an n-frame host sequence; frames outside that range are only
// uploaded as far as the temporal window needs them (the halo of a frame block).
static int sequence_impl(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                         int n, int k, int first, int count, void *const *host_out, bool out_u8,
                         int overlap, float *timings_ms)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(p && host_frames && host_out, "sequence_nlm: NULL argument");
    MID_REQUIRE(n >= 1 && k >= 0 && 2 * k + 2 <= kMaxFrames, "sequence_nlm: bad n=%d k=%d", n, k);
    MID_REQUIRE(first >= 0 && count >= 1 && first + count <= n, "sequence_nlm: bad range first=%d count=%d n=%d", first, count, n);
    MID_REQUIRE(p->width > 0 && p->height > 0, "sequence_nlm: bad size");
    const int f_lo = first - k < 0 ? 0 : first - k;                                  // first frame ever uploaded
    const int f_hi = first + count - 1 + k > n - 1 ? n - 1 : first + count - 1 + k;  // last one
    for (int i = f_lo; i <= f_hi; ++i) MID_REQUIRE(host_frames[i], "sequence_nlm: frame %d is NULL", i);
    for (int i = 0; i < count; ++i) MID_REQUIRE(host_out[i], "sequence_nlm: output %d is NULL", i);

    // MID_PIPE_GATED=1 selects the gated schedule above (same results bit for bit, measured no faster)
    {
        const char *e = getenv("MID_PIPE_GATED");
        if (overlap && e && e[0] == '1') return sequence_gated(ctx, p, host_frames, n, k, first, count, host_out, out_u8, timings_ms);
    }

    const size_t npix = (size_t)p->width * p->height;
    const size_t in_bytes = npix * (p->format == MID_FMT_RGBA8 ? 4 : 16);
    const size_t dl_bytes = npix * (out_u8 ? 4 : 16);            // one output frame, as it is written and downloaded
    const int n_up = f_hi - f_lo + 1;
    // Outputs can be filtered in batches of B frames per launch.  Measured on MI355X (16 x 1080p, 21x21/7x7):
    // B=1 2084 Mpixel/s, B=2 1341, B=4 1472, B=8 1403 -- coarser batches bunch the copies and lose overlap
    // while the kernel time barely changes, so one frame per launch is the default.
    int B = 1;
    if (const char *e = getenv("MID_PIPE_BATCH")) { B = atoi(e); if (B < 1) B = 1; }
    if (B > count) B = count;
    if (2 * k + 2 * B > kMaxFrames) B = (kMaxFrames - 2 * k) / 2;
    const int nb = (count + B - 1) / B;
    constexpr int DEPTH = 4;                                  // batches in flight per stage
    const int ring = n_up < 2 * k + DEPTH * B ? n_up : 2 * k + DEPTH * B;

    DeviceBufs dring, dout;
    if (int rc = dring.make(ring, in_bytes)) return rc;
    if (int rc = dout.make(DEPTH * B, dl_bytes)) return rc;
    EventPool up0, up1, c0, c1, d0, d1;
    for (EventPool *e : {&up0, &up1}) if (int rc = e->make(n_up)) return rc;
    for (EventPool *e : {&c0, &c1, &d0, &d1}) if (int rc = e->make(nb)) return rc;
    auto slot = [&](int f) { return dring.p[(f - f_lo) % ring]; };

    const auto wall0 = std::chrono::steady_clock::now();
    int next_upload = f_lo;
    auto upload = [&](int f) -> int {
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
        MID_HIP(hipMemcpyAsync(slot(f), host_frames[f], in_bytes, hipMemcpyHostToDevice, ctx->upload));
        MID_HIP(hipEventRecord(up1.ev[f - f_lo], ctx->upload));
        return MID_OK;
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
        if (bi >= DEPTH) MID_HIP(hipStreamWaitEvent(cs, d1.ev[bi - DEPTH], 0));

        const int lo = b0 - k < 0 ? 0 : b0 - k;
        const void *tbl[kMaxFrames];
        for (int f = lo; f <= need; ++f) tbl[f - lo] = slot(f);
        mid_pixel *o[kMaxFrames];
        for (int i = 0; i < bn; ++i) o[i] = (mid_pixel *)dout.p[(bi % DEPTH) * B + i];
        MID_HIP(hipEventRecord(c0.ev[bi], cs));
        // out_u8: GetImageFromGPU's u8 conversion (src/main.cpp:97-103) in the kernel's epilogue -- a quarter of the
        // bytes to write and to download
        if (int rc = nlm_temporal_out(ctx, p, tbl, need - lo + 1, k, b0 - lo, bn, (void *const *)o, out_u8 ? 1 : 0, cs)) return rc;
        MID_HIP(hipEventRecord(c1.ev[bi], cs));

        while (next_upload <= ahead) { if (int rc = upload(next_upload++)) return rc; }

        MID_HIP(hipStreamWaitEvent(ctx->download, c1.ev[bi], 0));
        MID_HIP(hipEventRecord(d0.ev[bi], ctx->download));
        for (int i = 0; i < bn; ++i)
            MID_HIP(hipMemcpyAsync(host_out[b0 - first + i], dout.p[(bi % DEPTH) * B + i], dl_bytes, hipMemcpyDeviceToHost, ctx->download));
        MID_HIP(hipEventRecord(d1.ev[bi], ctx->download));

        if (!overlap) {   // the reference's behaviour: a fence wait after every submit (src/main.cpp:1092)
            MID_HIP(hipStreamSynchronize(ctx->upload));
            MID_HIP(hipStreamSynchronize(ctx->compute));
            MID_HIP(hipStreamSynchronize(ctx->download));
        }
    }
    MID_HIP(hipStreamSynchronize(ctx->upload));
    MID_HIP(hipStreamSynchronize(ctx->compute));
    MID_HIP(hipStreamSynchronize(ctx->compute2));
    MID_HIP(hipStreamSynchronize(ctx->download));
    const auto wall1 = std::chrono::steady_clock::now();

    if (getenv("MID_PIPE_TRACE")) {   // development aid: per-batch stream timeline relative to the first upload
        for (int bi = 0; bi < nb; ++bi) {
            float k0, k1, e0, e1;
            (void)hipEventElapsedTime(&k0, up0.ev[0], c0.ev[bi]);  (void)hipEventElapsedTime(&k1, up0.ev[0], c1.ev[bi]);
            (void)hipEventElapsedTime(&e0, up0.ev[0], d0.ev[bi]);  (void)hipEventElapsedTime(&e1, up0.ev[0], d1.ev[bi]);
            fprintf(stderr, "batch %2d (%d frames)  compute %.3f-%.3f  down %.3f-%.3f ms\n", bi, B, k0, k1, e0, e1);
        }
    }
    if (timings_ms) {
        float kern = 0.f, copy = 0.f, ms = 0.f;
        for (int bi = 0; bi < nb; ++bi) {
            MID_HIP(hipEventElapsedTime(&ms, c0.ev[bi], c1.ev[bi])); kern += ms;
            MID_HIP(hipEventElapsedTime(&ms, d0.ev[bi], d1.ev[bi])); copy += ms;
        }
        for (int i = 0; i < n_up; ++i) { MID_HIP(hipEventElapsedTime(&ms, up0.ev[i], up1.ev[i])); copy += ms; }
        timings_ms[0] = std::chrono::duration<float, std::milli>(wall1 - wall0).count();
        timings_ms[1] = kern;
        timings_ms[2] = copy;
    }
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
    MID_REQUIRE(p && host_target && host_frames && host_out, "nlm_multiframe: NULL argument");
    MID_REQUIRE(n >= 1 && p->width > 0 && p->height > 0, "nlm_multiframe: bad n=%d or size", n);
    for (int i = 0; i < n; ++i) MID_REQUIRE(host_frames[i], "nlm_multiframe: frame %d is NULL", i);
    const size_t npix = (size_t)p->width * p->height;
    const size_t in_bytes = npix * (p->format == MID_FMT_RGBA8 ? 4 : 16), out_bytes = npix * 16;

    DeviceBufs dtarget, dslot, dW, dout;
    if (int rc = dtarget.make(1, in_bytes)) return rc;
    constexpr int SLOTS = 3;                                   // neighbour frames in flight: uploads run two dispatches ahead
    if (int rc = dslot.make(n < SLOTS ? n : SLOTS, in_bytes)) return rc;
    if (int rc = dW.make(1, npix * sizeof(mid_weightinfo))) return rc;
    if (int rc = dout.make(1, out_bytes)) return rc;
    EventPool up0, up1, c0, c1, misc;
    for (EventPool *e : {&up0, &up1, &c0, &c1})
        if (int rc = e->make(n)) return rc;
    if (int rc = misc.make(4)) return rc;

    const auto wall0 = std::chrono::steady_clock::now();
    // target + cleared weight buffer (the reference relies on a fresh allocation being zero)
    MID_HIP(hipEventRecord(misc.ev[0], ctx->upload));
    MID_HIP(hipMemcpyAsync(dtarget.p[0], host_target, in_bytes, hipMemcpyHostToDevice, ctx->upload));
    MID_HIP(hipEventRecord(misc.ev[1], ctx->upload));
    MID_HIP(hipMemsetAsync(dW.p[0], 0, npix * sizeof(mid_weightinfo), ctx->compute));
    MID_HIP(hipStreamWaitEvent(ctx->compute, misc.ev[1], 0));

    auto upload = [&](int f) -> int {
        if (f >= SLOTS) MID_HIP(hipStreamWaitEvent(ctx->upload, c1.ev[f - SLOTS], 0));   // slot still read by dispatch f-SLOTS
        MID_HIP(hipEventRecord(up0.ev[f], ctx->upload));
        MID_HIP(hipMemcpyAsync(dslot.p[f % SLOTS], host_frames[f], in_bytes, hipMemcpyHostToDevice, ctx->upload));
        MID_HIP(hipEventRecord(up1.ev[f], ctx->upload));
        return MID_OK;
    };
    int next_upload = 0;
    for (int f = 0; f < n; ++f) {
        // overlap: frames f+1, f+2 ride beside dispatch f (their slots were last read by dispatches already enqueued)
        const int ahead = overlap ? (f + SLOTS - 1 < n - 1 ? f + SLOTS - 1 : n - 1) : f;
        while (next_upload <= ahead) { if (int rc = upload(next_upload++)) return rc; }
        MID_HIP(hipStreamWaitEvent(ctx->compute, up1.ev[f], 0));
        MID_HIP(hipEventRecord(c0.ev[f], ctx->compute));
        if (int rc = mid_nlm_accum(ctx, p, dtarget.p[0], dslot.p[f % SLOTS], (mid_weightinfo *)dW.p[0], ctx->compute)) return rc;
        MID_HIP(hipEventRecord(c1.ev[f], ctx->compute));
        if (!overlap) {   // fence after every submit, src/main.cpp:1092
            MID_HIP(hipStreamSynchronize(ctx->compute));
            MID_HIP(hipStreamSynchronize(ctx->upload));
        }
    }
    mid_normalize_params np{p->width, p->height};
    if (int rc = mid_normalize(ctx, &np, (const mid_weightinfo *)dW.p[0], (mid_pixel *)dout.p[0], ctx->compute)) return rc;
    MID_HIP(hipEventRecord(misc.ev[2], ctx->compute));
    MID_HIP(hipMemcpyAsync(host_out, dout.p[0], out_bytes, hipMemcpyDeviceToHost, ctx->compute));
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
