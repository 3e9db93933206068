/*
 * mi_denoise.h -- C-ABI of libmi_denoise.so, the MI355X (gfx950) denoise hot path.
 *
 * The reference (Reefufui/image_denoising_filter) has no library/FFI surface: its "operator
 * interface" is the set of descriptor bindings + push-constant blocks each GLSL kernel is
 * dispatched with, and the ComputeApplication methods that drive them.  Each entry point
 * below names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 (MID_OK) or a MID_ERR_* code; mid_last_error() gives the
 *     thread-local message.  No exception crosses the ABI.  (Reference: VK_CHECK_RESULT =
 *     print+assert, src/vk_utils.h:55-63; std::runtime_error -> EXIT_FAILURE, src/main.cpp:1987.)
 *   - all image pointers are DEVICE pointers unless a name says host; the caller owns every
 *     buffer (reference: the application owns every VkBuffer/VkImage, src/main.cpp:1398-1437).
 *   - `stream` is a hipStream_t passed as void*; NULL = the context's own compute stream.
 *     Calls are asynchronous on that stream.  (The legacy default stream's handle is 0 too, so it cannot be named here:
 *     a caller whose buffers are produced on the default stream -- e.g. torch.cuda.current_stream().cuda_stream == 0 --
 *     passes a created stream that it orders after that work, or synchronises first.)
 *   - images are row-major, pixel index = width*y + x (shaders/bialteral.comp:81).
 *   - there is NO CPU fallback: without a usable HIP device mid_ctx_create fails.
 */
#ifndef MI_DENOISE_H
#define MI_DENOISE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MID_VERSION 100

enum {
    MID_OK = 0,
    MID_ERR_INVALID = 1,      /* bad argument / illegal mode combination (reference: assert, src/main.cpp:1315-1316) */
    MID_ERR_HIP = 2,          /* a HIP runtime call failed */
    MID_ERR_NO_DEVICE = 3,    /* no gfx950-capable device: the product path refuses to run */
    MID_ERR_UNSUPPORTED = 4,  /* parameter combination without a kernel */
    MID_ERR_IO = 5            /* file decode/encode failure (codec + CLI layer) */
};

/* struct Pixel {float r,g,b,a;}  -- src/main.cpp:39-41, the shaders' `struct Pixel{vec4 value;}` */
typedef struct mid_pixel { float r, g, b, a; } mid_pixel;

/* struct WeightInfo {vec4 weightColor; float normWeight;} at std430 stride 32 B --
 * shaders/nonlocal.comp:10-14, shaders/bialteral_layers.comp:8-12, shaders/normalize.comp:13-17;
 * host agrees: src/main.cpp:43-46 (struct NLM), :1399 (bufferSizeWeights). */
typedef struct mid_weightinfo {
    float weightColor[4];
    float normWeight;
    float _pad[3];
} mid_weightinfo;

/* Texel format of an input image: the reference creates RGBA32F textures for .exr and
 * RGBA8_UNORM for .png (src/texture.cpp:16, src/main.cpp:333-346). */
enum { MID_FMT_RGBA32F = 0, MID_FMT_RGBA8 = 1 };

/* Addressing of the bilateral input: bialteral.comp (sampler2D, 2-D texelFetch, out-of-image
 * texel = 0) vs bialteral_linear.comp (samplerBuffer, flat index c + j + i*width: columns wrap
 * into the adjacent row, index outside [0,N) = 0).  m_linear = !nonlinear, src/main.cpp:1311. */
enum { MID_LAYOUT_TEXTURE = 0, MID_LAYOUT_LINEAR = 1 };

/* Push-constant block of bialteral.comp / bialteral_linear.comp / bialteral_layers.comp
 * (shaders/bialteral.comp:13-20: {int width; int height; float spatialSigma; float colorSigma},
 * 16 B, pushed at src/main.cpp:801-808 and :873-877 with {2.0f, 0.2f}); the first 16 bytes are
 * that block byte for byte.  `radius` replaces `#define TEXEL_WINDOW 20` (shaders/bialteral.comp:5). */
typedef struct mid_bilateral_params {
    int32_t width;
    int32_t height;
    float   spatialSigma;
    float   colorSigma;
    int32_t radius;   /* window is (2*radius+1)^2; kernels exist for 1..24 */
    int32_t layout;   /* MID_LAYOUT_* (ignored by the layers kernels: always texture) */
    int32_t format;   /* MID_FMT_* of `in` */
} mid_bilateral_params;

/* Push-constant block of nonlocal.comp (shaders/nonlocal.comp:16-22: {int width; int height;
 * float filteringParameter}, 12 B, pushed at src/main.cpp:865-872 with 0.5f).  The ranges
 * replace `#define WINDOW 7` / `#define PATCH_WINDOW 3` (shaders/nonlocal.comp:5-6) and are
 * HALF-OPEN [lo,hi) like the shader's loops (:36-44): the reference is search [-7,7) patch
 * [-3,3); the 21x21 / 7x7 benchmark configuration is search [-10,11) patch [-3,4).
 * Limits: both ranges contain 0, search width <= 64, patch width <= 16.  Every search range and every patch of the forms
 * [-P,P) and [-P,P] up to 16 wide runs on the LDS-tiled kernel (as long as the tile fits 160 KB: search width <= 51 at a
 * 7x7 patch); anything else (lopsided patches, wider windows) is computed by a per-pixel kernel -- same results, the
 * reference's S^2 * P^2 work. */
typedef struct mid_nlm_params {
    int32_t width;
    int32_t height;
    float   filteringParameter;
    int32_t search_lo, search_hi;
    int32_t patch_lo, patch_hi;
    int32_t format;   /* MID_FMT_* of target and neighbour images */
} mid_nlm_params;

/* Push-constant block of normalize.comp (shaders/normalize.comp:19-24), 8 B. */
typedef struct mid_normalize_params { int32_t width; int32_t height; } mid_normalize_params;

typedef struct mid_ctx mid_ctx;   /* opaque; bound to one HIP device */

/* ---- context / errors / memory -------------------------------------------------------
 * Replaces the Vulkan bootstrap (src/vk_utils.cpp:13-305) and buffer factories
 * (src/main.cpp:247-401): a context is one device + one compute stream. */
int         mid_ctx_create(int device, mid_ctx **out);
void        mid_ctx_destroy(mid_ctx *ctx);
/* Frees the device buffers and events the frame pipeline (mid_sequence_nlm*, mid_nlm_multiframe) keeps in the context
 * between calls -- at 1080p RGBA32F and k = 2 about 400 MB -- and the 32 MiB of page-locked bounce buffers; the next call
 * allocates again.  Without it the cache follows the calls: it grows to the largest frame size seen, and a call whose frames
 * are more than four times smaller than the cached buffers gives those back and keeps buffers of its own size. */
int         mid_ctx_release_cached(mid_ctx *ctx);
const char *mid_last_error(void);
int         mid_version(void);
int         mid_device_name(mid_ctx *ctx, char *buf, size_t buflen);
/* The context's four streams -- [0] compute (the default for stream = NULL), [1] the frame pipeline's second kernel stream,
 * [2] its upload and [3] its download stream -- are created together by mid_ctx_create; the two copy streams at the device's
 * highest stream priority (numerically lowest: `greatest`), so that they never share one of the runtime's hardware queues with
 * the kernel streams or with the caller's streams (csrc/capi.cpp has the measurements).  Stands where the reference takes its
 * single queue (vkGetDeviceQueue, src/main.cpp:1335). */
int         mid_ctx_stream_priorities(mid_ctx *ctx, int priority[4], int *least, int *greatest);

int mid_alloc(mid_ctx *ctx, size_t bytes, void **dptr);              /* CreateWriteOnlyBuffer & co */
int mid_free(mid_ctx *ctx, void *dptr);
int mid_alloc_host(mid_ctx *ctx, size_t bytes, void **hptr);         /* pinned staging (CreateStagingBuffer/CreateDynamicBuffer) */
int mid_free_host(mid_ctx *ctx, void *hptr);
/* Pin memory the caller already owns (a decoded frame in a std::vector, say) so that copies from/to it are truly
 * asynchronous (the pinning holds for every device of the process); undo with mid_host_unregister before the memory is
 * freed and after every copy that uses it has completed.  Measured on MI355X, 16 x 1080p RGBA32F through
 * mid_sequence_nlm: buffers from mid_alloc_host 12.8 ms, memory registered in place 25.1 ms (+6 ms to register), pageable
 * memory (bounced inside the library, see mid_memcpy_h2d) slower than either -- so decode into mid_alloc_host buffers when
 * throughput matters, and register in place when the frames already exist in ordinary memory. */
int mid_host_register(mid_ctx *ctx, void *hptr, size_t bytes);
int mid_host_unregister(mid_ctx *ctx, void *hptr);
/* Host <-> device copies.  Page-locked host memory (mid_alloc_host, mid_host_register, mid_image_load_pinned) is DMA'd in
 * place and the call is asynchronous on `stream`.  ANY OTHER host memory is accepted too, but it never reaches the HIP
 * runtime (whose pin-on-the-fly path for pageable copies of more than 1 MiB the library does not rely on): it is moved in
 * 8 MiB chunks through page-locked bounce buffers the context owns (32 MiB, allocated on first use), and the call returns
 * when src_host has been consumed (h2d; the last chunks may still be in flight on `stream`) or dst_host holds the data (d2h).
 * The same holds for every host frame handed to mid_sequence_nlm* / mid_nlm_multiframe.  Pageable copies cost a host memcpy
 * on top of the DMA (measured: see LABNOTES R5.1), so decode into pinned memory when throughput matters.
 * Threads: a context has ONE pair of bounce buffers per direction, guarded by a lock the copying thread holds from its first chunk
 * to its last, INCLUDING the waits for `stream` to reach each chunk.  Pageable copies of one direction on one context are therefore
 * serialised, also across independent streams, and two threads must not make them depend on each other: if thread A's copy sits on a
 * stream that waits (hipStreamWaitEvent) for work queued behind thread B's pageable copy of the same direction on the same context,
 * A holds the lock while its stream waits for B and B waits for the lock -- a deadlock.  Pinned copies take no lock; threads that
 * need independent pageable copies use one context each (contexts share nothing). */
int mid_memcpy_h2d(mid_ctx *ctx, void *dst, const void *src_host, size_t bytes, void *stream);  /* LoadImageDataToBuffer + copy-to-texture, src/main.cpp:1105-1142,990-1076 */
int mid_memcpy_d2h(mid_ctx *ctx, void *dst_host, const void *src, size_t bytes, void *stream);  /* vkCmdCopyBuffer to staging + GetImageFromGPU, src/main.cpp:835-840,91-123 */
int mid_memset(mid_ctx *ctx, void *dst, int value, size_t bytes, void *stream);                 /* the reference never clears its weight buffer; callers of *_accum must */
int mid_stream_sync(mid_ctx *ctx, void *stream);                     /* vkWaitForFences, src/main.cpp:1092 */

/* ---- recorded command sequences ---------------------------------------------------------
 * The reference works on RECORDED command buffers: every RecordCommandsOf* brackets its dispatches, barriers and copies with
 * vkBeginCommandBuffer / vkEndCommandBuffer (src/main.cpp:791/846, 855/886, 895/988, 996/1075) and RunCommandBuffer submits the
 * recording (vkQueueSubmit, :1078-1103); it records anew before each submission.  Here a sequence is recorded ONCE and submitted
 * as often as needed -- the HIP counterpart of a command buffer is a captured graph (csrc/recording.cpp):
 *   mid_record_begin(ctx, stream)          vkBeginCommandBuffer: `stream` (NULL = the context's compute stream) stops executing
 *                                          and records instead;
 *   ... kernel-level calls on that stream  mid_bilateral*, mid_nlm_accum, mid_nlm_temporal, mid_normalize, mid_pack_u8,
 *                                          mid_unpack_u8, mid_memset, mid_memcpy_* from / to PAGE-LOCKED memory: everything that only
 *                                          enqueues on the one stream.  Arguments are taken as they are at record time: the same
 *                                          buffers are read and written by every submission (like bound descriptors), their
 *                                          CONTENT is whatever it is when the submission runs;
 *   mid_record_end(ctx, stream, &rec)      vkEndCommandBuffer; the stream executes again;
 *   mid_recording_submit(rec, stream)      vkQueueSubmit: asynchronous on `stream` (any stream of the context's device), ordered
 *                                          like one launch; the same recording must not be in flight twice at the same time;
 *   mid_recording_destroy(rec)             after its last submission has completed.
 * One runtime call per submission instead of one per dispatch.  Measured on MI355X (profiles/r06_recording_replay.txt, bench.py
 * also.graph_replay_literal_nlm): the same bytes, and NOT faster than the calls issued one by one from compiled code -- about 4 us per
 * launch either way, and the reference's literal multi-frame sequence is bound by its chain of dependent kernels at every frame size.
 * A matter of program shape; for speed use the fused entry points (mid_nlm_temporal, mid_bilateral_batch, mid_bilateral_layers).
 * mid_record_begin and mid_record_end of one recording are called on the same thread.
 * Refused while a stream records (MID_ERR_INVALID, the recording stays valid): calls that wait on the host, drive several streams or
 * bounce pageable memory -- mid_stream_sync, mid_timer_tick/tock, mid_sequence_nlm*, mid_nlm_multiframe, mid_nlm_temporal_sharded,
 * mid_comm_loopback, copies from / to pageable host memory.  Do not allocate or free (mid_alloc*, mid_free*, mid_host_register) on the
 * recording thread between begin and end: the runtime fails such calls and invalidates the recording (mid_record_end then reports
 * it).  The capture is thread-local: other threads keep using their own streams meanwhile.  A kernel's first ever launch may be
 * inside a recording (raising its LDS limit is not a stream operation). */
typedef struct mid_recording mid_recording;
int mid_record_begin(mid_ctx *ctx, void *stream);
int mid_record_end(mid_ctx *ctx, void *stream, mid_recording **out);
int mid_recording_submit(mid_recording *rec, void *stream);
int mid_recording_info(mid_recording *rec, int *n_nodes /* graph nodes */, int *n_kernels /* of them kernel launches */);
int mid_recording_destroy(mid_recording *rec);

/* ---- a1/a2: plain bilateral -----------------------------------------------------------
 * Replaces the dispatch of shaders/bialteral.comp / bialteral_linear.comp recorded by
 * RecordCommandsOfExecuteAndTransfer(normKernel=false), src/main.cpp:785-847: bindings
 * {0: vec4 out[N]; 1: image}. */
int mid_bilateral(mid_ctx *ctx, const mid_bilateral_params *p,
                  const void *in, mid_pixel *out, void *stream);
/* The same dispatch for n_frames independent frames of one size in ONE launch (grid = tiles x frames): what the
 * reference does by calling RunOnGPU once per file (src/main.cpp:1952-1985) in one launch instead of n (measured: the same
 * rate per frame -- at r = 8 a 1080p frame is 2040 workgroups on 1024 slots, four 40 KB tiles per CU, 1.99 rounds, so there is
 * little tail to save; tools/microbench19.hip).  in/out: host arrays of n_frames device pointers.
 * Results are bit-identical to n_frames calls of mid_bilateral (same tile code).
 * No aliasing: an `out` buffer must not be any `in` buffer of the same call (all frames are filtered concurrently) nor
 * appear twice; mid_bilateral likewise rejects in == out.  Violations return MID_ERR_INVALID. */
int mid_bilateral_batch(mid_ctx *ctx, const mid_bilateral_params *p, const void *const *in /* n_frames device ptrs */,
                        mid_pixel *const *out /* n_frames device ptrs */, int n_frames, void *stream);

/* ---- a3: layer-guided bilateral ---------------------------------------------------------
 * mid_bilateral_layers_accum = one dispatch of shaders/bialteral_layers.comp
 * (RecordCommandsOfExecuteNLM(nlm=false), src/main.cpp:849-887): W[p] += sums, bindings
 * {0: WeightInfo W[N]; 1: inputTex; 2: layerTex (always RGBA8, src/main.cpp:1396,1419-1420)}.
 * mid_bilateral_layers = the whole per-layer loop src/main.cpp:1610-1623 plus normalize
 * (:1649-1652) fused in one kernel: no WeightInfo traffic. */
int mid_bilateral_layers_accum(mid_ctx *ctx, const mid_bilateral_params *p, const void *in,
                               const uint32_t *layer_rgba8, mid_weightinfo *W, void *stream);
int mid_bilateral_layers(mid_ctx *ctx, const mid_bilateral_params *p, const void *in,
                         const uint32_t *const *layers_rgba8 /* host array of device ptrs */,
                         int n_layers, mid_pixel *out, void *stream);

/* ---- a4: non-local means ----------------------------------------------------------------
 * mid_nlm_accum = one dispatch of shaders/nonlocal.comp (RecordCommandsOfExecuteNLM(nlm=true),
 * src/main.cpp:849-887, loop :1577-1606): W[p] += (weightColor, 0.001 + sum of weights) for one
 * neighbour frame, target fixed; bindings {0: W; 1: u_targetImage; 2: u_neighbourImage}.
 * mid_nlm_temporal = the multi-frame mode (src/main.cpp:1539-1606) for an animation: for each
 * output frame t in [first, first+count) it accumulates over neighbour frames
 * max(0,t-k)..min(n_frames-1,t+k) in ascending order (target = frame t) and normalizes, all in
 * one launch (grid.z = output frame); k = 0 is independent single-frame NLM over a batch. */
int mid_nlm_accum(mid_ctx *ctx, const mid_nlm_params *p, const void *target,
                  const void *neighbour, mid_weightinfo *W, void *stream);
/* (mid_nlm_temporal: no aliasing -- an `out` buffer must not be a frame of the sequence nor appear twice: MID_ERR_INVALID, as for mid_bilateral_batch.) */
int mid_nlm_temporal(mid_ctx *ctx, const mid_nlm_params *p,
                     const void *const *frames /* host array of n_frames device ptrs */,
                     int n_frames, int k, int first, int count,
                     mid_pixel *const *out /* host array of `count` device ptrs */, void *stream);

/* ---- a5: normalize ----------------------------------------------------------------------
 * One dispatch of shaders/normalize.comp (RecordCommandsOfExecuteAndTransfer(normKernel=true)):
 * out = weightColor / normWeight, or (1,0,1,1) where normWeight == 0; bindings {0: out; 1: W}. */
int mid_normalize(mid_ctx *ctx, const mid_normalize_params *p,
                  const mid_weightinfo *W, mid_pixel *out, void *stream);

/* ---- a6: u8 <-> float -------------------------------------------------------------------
 * unpack flavour 0 = UNORM texel decode c/255 (src/texture.cpp:16, src/main.cpp:341);
 *        flavour 1 = CPU decode (float)c * (1.0f/255.0f) (src/main.cpp:1804-1807).
 * pack = (unsigned char)(255.0f*v), truncation (GetImageFromGPU, src/main.cpp:97-103), clamped
 * to [0,255] only where that cast is undefined in C.  n_values counts channels, not pixels.
 * The u8 buffer must be 4-byte aligned and the float buffer 16-byte aligned (hipMalloc'd buffers are). */
int mid_unpack_u8(mid_ctx *ctx, const uint8_t *in, size_t n_values, int flavour, float *out, void *stream);
int mid_pack_u8(mid_ctx *ctx, const float *in, size_t n_values, uint8_t *out, void *stream);

/* ---- a8: frame pipeline -----------------------------------------------------------------
 * Replaces RecordCommandsOfOverlappingNLM + the ping-pong loop (src/main.cpp:889-989,
 * :1539-1573): host frames in, host frames out.  Frame t+k+1 is uploaded (hipMemcpyAsync from
 * pinned slots on an upload stream) while frame t is filtered on the compute stream and frame
 * t-1 is downloaded on a third stream; a device ring keeps 2k+4 frames and four
 * output slots in flight, so each stage may run up to three frames away from its neighbours, and
 * consecutive frames are filtered on two alternating kernel streams (one launch's tail overlaps the next one's head).
 * host_frames/host_out are arrays of n_frames HOST pointers (RGBA32F or RGBA8 per p->format;
 * output always RGBA32F).  Synchronous: returns when every output is on the host.
 * The device ring, the output slots and the events are kept in the context between calls (grown when a call needs more
 * or larger ones; mid_ctx_release_cached / mid_ctx_destroy free them), so only a context's first call -- or the first
 * at a larger frame size -- allocates.  Calls on one context are serialised (they share its four streams).
 * timings_ms (optional, 3 floats): wall time of the WHOLE call, set-up included; sum of kernel time; sum of copy time. */
int mid_sequence_nlm(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                     int n_frames, int k, mid_pixel *const *host_out, int overlap, float *timings_ms);
/* Same, for the output frames [first, first+count) only (host_out has `count` entries): the unit of
 * frame-block sharding -- a device filters its block and uploads the k halo frames on either side. */
int mid_sequence_nlm_range(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                           int n_frames, int k, int first, int count, mid_pixel *const *host_out,
                           int overlap, float *timings_ms);
/* Same, with the reference's u8 read-back conversion (GetImageFromGPU, src/main.cpp:97-103: truncating
 * (unsigned char)(255.0f*v), see mid_pack_u8) done on the device before the download: host_out receives RGBA8
 * frames, a quarter of the PCIe bytes -- the LDR (PNG in, PNG out) path of the reference. */
int mid_sequence_nlm_range_u8(mid_ctx *ctx, const mid_nlm_params *p, const void *const *host_frames,
                              int n_frames, int k, int first, int count, uint8_t *const *host_out,
                              int overlap, float *timings_ms);

/* Debug export: the DEVICE timeline of this context's last mid_sequence_nlm* call, from the events the call recorded on
 * its streams (no profiler: the call ran at its own pace).  All times in ms from the start of the call's first upload.
 * upload_ms[2*i], [2*i+1]: start / end of the upload of frame first_upload_frame + i; output_ms[4*j .. 4*j+3]: kernel
 * start, kernel end, download start, download end of output frame first_output_frame + j (output j runs on kernel stream
 * j & 1).  cap = rows either array can hold.  Valid until the next pipeline call or mid_ctx_release_cached on the context.
 * Stands where the reference prints its per-submit timestamps (src/main.cpp:1095-1101). */
int mid_pipe_last_timeline(mid_ctx *ctx, int cap, float *upload_ms /* cap x 2 */, int *n_uploads, int *first_upload_frame,
                           float *output_ms /* cap x 4 */, int *n_outputs, int *first_output_frame);

/* The reference's literal multi-frame mode (src/main.cpp:1539-1606): ONE target, its neighbour frames
 * streamed from the host.  W = sum over frames of one nonlocal.comp dispatch each (target fixed), then
 * normalize.  overlap != 0 replaces RecordCommandsOfOverlappingNLM (:889-989): frame i+1 is uploaded on
 * the upload stream into the other of two device slots while frame i is being accumulated.
 * host_frames: n_frames HOST pointers (format p->format); host_out: w*h RGBA32F on the host.
 * timings_ms (optional, 3 floats): wall, kernel sum, copy sum. */
int mid_nlm_multiframe(mid_ctx *ctx, const mid_nlm_params *p, const void *host_target,
                       const void *const *host_frames, int n_frames, mid_pixel *host_out,
                       int overlap, float *timings_ms);

/* ---- 8e: an animation sharded over GPUs (frame blocks + RCCL halo) ----------------------------
 * New capability with the reference's multi-frame semantics (the reference is single-device: deviceId{0},
 * src/main.cpp:1321; its neighbour-frame loop is src/main.cpp:1577-1606).  One rank per GPU owns a contiguous block of
 * an n-frame sequence, resident in its HBM; temporal NLM over t-k..t+k needs the k frames on either side of the block,
 * which are exchanged GPU to GPU in ONE step (ncclSend/ncclRecv in one group, point to point over xGMI) -- no other
 * collective.  Interior outputs are launched while the halo is in flight, boundary outputs after it.
 * RCCL is loaded at run time on the first mid_comm_* call (no link-time dependency; MID_ERR_UNSUPPORTED if absent).
 *
 * mid_shard_block: the partition -- rank r owns frames [start, start+count), the first n % world ranks one extra.
 * mid_shard_halo_plan / mid_shard_launch_plan: the exchange and the launches of one rank as pure data (host only, no
 *   GPU): receives/sends as (peer rank, global frame id) in issue order; launch rows {interior, w_lo, w_hi, first,
 *   count, out_offset}: frames w_lo..w_hi form the table handed to mid_nlm_temporal, outputs are table entries
 *   [first, first+count), stored at block-relative out_offset.  `cap` = capacity of the caller's arrays.
 * mid_comm_unique_id + mid_comm_create: one process per GPU (rank 0 makes the id and hands it to the others out of
 *   band -- a file, MPI, torch.distributed); mid_comm_create_all: one process, one context per device.
 *   RCCL is looked up as librccl.so.1 / librccl.so / /opt/rocm/lib/librccl.so.1; the environment variable
 *   MID_RCCL_LIBRARY (read once, on the first mid_comm_* call) names the library to load instead.
 * mid_nlm_temporal_sharded: `block` = this rank's `count` device frames in order, `out` = `count` device outputs.
 *   Asynchronous on `stream`.  COLLECTIVE: every rank of the communicator must call it, with the same n_frames, k and
 *   frame size -- also a rank that owns no frame.  Everything that can fail locally (arguments, the stream rule below,
 *   receive-buffer allocation) is checked before the first RCCL call, so a rank that returns an error there has not
 *   entered the exchange; its peers then wait for it, and the caller must mid_comm_abort (or destroy) the communicator
 *   on EVERY rank -- there is no other way out of a half-entered collective.
 *   Stream rule: the receive buffers are reused from call to call, ordered only through the stream the calls are issued
 *   on.  A call on a different stream than the previous one returns MID_ERR_INVALID unless the previous call's work has
 *   finished (e.g. after mid_stream_sync on the earlier stream).
 *   Results are the bits of one mid_nlm_temporal over the whole sequence by construction (same kernels, same tile
 *   shape, same frame tables: tests/test_shard_native_plan.py).  Executed so far: with RCCL itself for world = 1; with 2-4
 *   ranks sharing one device against a test-only stand-in for the transport (tests/standin_rccl), bit-identical in 12
 *   partitions.  RCCL between two devices has not run yet: bench.py gates it with `bit_identical_to_single_launch_per_rank`.
 * mid_comm_reserve: allocates the 2k receive buffers for frames of up to max_frame_bytes up front, so that no sharded
 *   call allocates (optional; otherwise they are allocated on first use and retired, not freed, when the frame size grows).
 * mid_comm_abort: ncclCommAbort -- tears the rank's connections down without waiting for outstanding operations; the
 *   handle then only accepts mid_comm_destroy.
 * mid_comm_loopback: a send+receive addressed to this very rank (the exchange's call pattern without the wire), asynchronous on
 *   `stream`; mid_comm_last_loopback: (waits for it) t[0] start / t[1] end of that group on the exchange stream, in ms from the
 *   moment `stream` reached the call.  Its events are its own: the last sharded call's timeline is not disturbed.
 * mid_comm_last_exchange: bytes received/sent by the last sharded call and (waits for it) the exchange's duration.
 * mid_comm_last_timeline: (waits for the call) its device timeline in ms from the call's first event on the caller's stream:
 *   t[0] exchange start, t[1] exchange end (both 0 when nothing was exchanged), t[2] end of the interior launches, t[3] end of
 *   the last launch.  The share of the exchange hidden behind interior compute is (min(t[1], t[2]) - t[0]) / (t[1] - t[0]).
 * mid_comm_last_issue_order: what the call put on its streams, in host issue order -- 'X' exchange group (exchange stream), 'I'
 *   interior launch (caller's stream), then per edge of the block 'W' the wait for the exchange and 'B' the boundary launch, on
 *   that edge's boundary stream of the communicator ("X I.. W B W B"), so that boundary workgroups fill the interior launch's
 *   tail; the caller's stream joins both before the call's work is complete on it; every 'I' precedes the first 'W' by construction.
 * mid_comm_stream_priority: the exchange stream's priority and the device's range (numerically lower = higher); the stream
 *   is created with the highest, so RCCL's send/receive kernels are dispatched ahead of queued interior workgroups.
 * mid_comm_boundary_priority: the two boundary streams' priority -- the device's LOWEST (`least`).  Streams and hardware queues:
 *   the HIP runtime gives each of its three priority levels a pool of four hardware queues and a queue runs its packets in order,
 *   so two busy streams that share a queue serialise.  The library's streams are placed so that this cannot happen among them,
 *   whatever the application creates: highest level = upload, download, exchange (3 of 4 queues); lowest level = the two boundary
 *   streams (2 of 4); default level = the context's two kernel streams, beside the caller's own streams. */
typedef struct mid_comm mid_comm;
#define MID_COMM_ID_BYTES 128
int mid_shard_block(int n_frames, int world, int rank, int *start, int *count);
int mid_shard_halo_plan(int n_frames, int world, int k, int rank, int cap,
                        int *recv_peer, int *recv_frame, int *n_recv, int *send_peer, int *send_frame, int *n_send);
int mid_shard_launch_plan(int n_frames, int world, int k, int rank, int cap, int *rows /* cap x 6 ints */, int *n_rows);
int mid_comm_unique_id(uint8_t id[MID_COMM_ID_BYTES]);
int mid_comm_create(mid_ctx *ctx, const uint8_t id[MID_COMM_ID_BYTES], int rank, int world, mid_comm **out);
int mid_comm_create_all(mid_ctx *const *ctxs, int world, mid_comm **out /* world entries */);
int mid_comm_destroy(mid_comm *comm);
int mid_comm_abort(mid_comm *comm);
int mid_comm_reserve(mid_comm *comm, size_t max_frame_bytes, int k);
int mid_comm_rank(mid_comm *comm, int *rank, int *world);
int mid_comm_loopback(mid_comm *comm, const void *src, void *dst, size_t bytes, void *stream);
int mid_comm_last_loopback(mid_comm *comm, float t_ms[2]);
int mid_nlm_temporal_sharded(mid_comm *comm, const mid_nlm_params *p, const void *const *block /* count device frames */,
                             int n_frames, int k, mid_pixel *const *out /* count device frames */, void *stream);
int mid_comm_last_exchange(mid_comm *comm, size_t *bytes_recv, size_t *bytes_sent, float *exchange_ms);
int mid_comm_last_timeline(mid_comm *comm, float t_ms[4]);
int mid_comm_last_issue_order(mid_comm *comm, char *buf, size_t buflen);
int mid_comm_stream_priority(mid_comm *comm, int *priority, int *least, int *greatest);
int mid_comm_boundary_priority(mid_comm *comm, int priority[2]);
/* What RCCL itself reports for the communicator: ncclCommCount, ncclCommUserRank, ncclGetVersion (-1 where unavailable). */
int mid_comm_rccl_info(mid_comm *comm, int *nranks, int *user_rank, int *version);

/* ---- 8f-2: image files ------------------------------------------------------------------
 * mid_image_load = LoadImages (src/main.cpp:145-229): ".exr" -> RGBA32F (tinyexr LoadEXR: missing
 * alpha = 1), anything else is decoded as PNG -> RGBA8 (lodepng::decode).  `data` is host memory
 * owned by the library until mid_image_free.  mid_image_save = SaveEXR(rgba,w,h,4,0) (:1699) for
 * MID_FMT_RGBA32F, lodepng::encode (:1717) for MID_FMT_RGBA8.  Host-only: no GPU needed. */
typedef struct mid_image {
    int32_t width, height;
    int32_t format;   /* MID_FMT_* */
    void   *data;
} mid_image;
int  mid_image_load(const char *path, mid_image *out);
void mid_image_free(mid_image *img);
/* The same decode, but the pixels are written DIRECTLY into pinned (page-locked) host memory of `ctx`'s device, so the
 * frame can be handed to mid_sequence_nlm* / mid_memcpy_h2d as a true asynchronous DMA source without another copy --
 * the reference likewise decodes and memcpy's straight into its mapped staging buffer (LoadImageDataToBuffer,
 * src/main.cpp:1105-1142).  Release with mid_image_free_pinned (never mid_image_free). */
int  mid_image_load_pinned(mid_ctx *ctx, const char *path, mid_image *out);
int  mid_image_free_pinned(mid_ctx *ctx, mid_image *img);
int  mid_image_save(const char *path, const void *data, int32_t width, int32_t height, int32_t format);
/* The codecs work on the independent blocks of a file in parallel (EXR chunks; PNG: filter rows and 1 MiB deflate segments on
 * encode -- the inflate of a PNG is one serial stream): by default on up to min(16, hardware threads) host threads per call.
 * mid_image_threads(n) sets that number FOR IMAGE CALLS MADE BY THE CALLING THREAD (n = 0: the default again, n < 0: query only)
 * and returns the previous setting.  A host that loads or saves many files at once -- one per thread -- sets 1 on those threads:
 * mi_denoise --animation decodes and encodes its frames that way (64 x 1080p PNG: the serial inflate of one file no longer
 * serialises the sequence).  File bytes do not depend on the thread count.  The reference does its image I/O on one thread
 * (lodepng / tinyexr calls, src/main.cpp:155,196,1699,1717). */
int  mid_image_threads(int n);

/* ---- measurement helper -----------------------------------------------------------------
 * Timestamps around a region on a stream (reference: vkCmdWriteTimestamp pool,
 * src/main.cpp:747-755,793-796,812-814).  tick/tock record hipEvents on `stream`;
 * mid_timer_ms waits for the tock and returns elapsed milliseconds. */
typedef struct mid_timer mid_timer;
int mid_timer_create(mid_ctx *ctx, mid_timer **out);
int mid_timer_destroy(mid_timer *t);
int mid_timer_tick(mid_timer *t, void *stream);
int mid_timer_tock(mid_timer *t, void *stream);
int mid_timer_ms(mid_timer *t, float *ms);

/* Named host ranges for a trace (ROCTx; `rocprofv3 --marker-trace`): the counterpart, for a profiler's timeline, of the
 * timestamp pairs the reference writes around every submit (src/main.cpp:793-796,812-814,842-844).  The frame pipeline
 * emits "upload f" / "nlm t" / "download t" inside one range per call; callers bracket their own stages with these two.
 * Bound to whatever ROCTx the process has ALREADY loaded (rocprofv3 preloads one); nothing is loaded otherwise and the
 * calls do nothing.  Return 1 when a ROCTx is present, 0 when not. */
int mid_range_push(const char *name);
int mid_range_pop(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_DENOISE_H */
