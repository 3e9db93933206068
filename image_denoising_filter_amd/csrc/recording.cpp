// recording.cpp -- record a sequence of kernel-level calls once, submit it many times (mid_record_* / mid_recording_*).
// The reference works on recorded command buffers: vkBeginCommandBuffer ... vkEndCommandBuffer inside every RecordCommandsOf*
// (src/main.cpp:791/846, 855/886, 895/988, 996/1075), submitted by RunCommandBuffer (vkQueueSubmit + vkWaitForFences, :1078-1103)
// -- and it records anew before every submission.  The HIP counterpart of a recorded command buffer is a captured graph:
// mid_record_begin puts the stream into capture mode, the kernel-level entry points of this library enqueue onto it as they always
// do (they only launch kernels / async copies on the stream they are given: no allocation, no host-side wait, no second stream),
// mid_record_end instantiates the graph, mid_recording_submit launches it: one runtime call per sequence instead of one per
// dispatch.  Measured (profiles/r06_recording_replay.txt): same bytes, no faster than a stream of launches from compiled code on
// this runtime (about 4 us per launch either way) -- the API is there for the program shape the reference has, not as a speed-up.
// The capture is thread-local (hipStreamCaptureModeThreadLocal): other threads of the process keep working on their own streams.
#include "common.hpp"
#include <new>

struct mid_recording {
    mid_ctx *ctx;
    hipGraph_t graph;
    hipGraphExec_t exec;
    int n_nodes, n_kernels;
};

using namespace mid;

extern "C" int mid_record_begin(mid_ctx *ctx, void *stream)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    MID_HIP(hipStreamIsCapturing(b.s, &st));
    MID_REQUIRE(st == hipStreamCaptureStatusNone, "record_begin: the stream is already recording");
    MID_HIP(hipStreamBeginCapture(b.s, hipStreamCaptureModeThreadLocal));
    return MID_OK;
}

extern "C" int mid_record_end(mid_ctx *ctx, void *stream, mid_recording **out)
{
    Bind b(ctx, stream);
    if (b.rc) return b.rc;
    MID_REQUIRE(out != nullptr, "record_end: out is NULL");
    *out = nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    MID_HIP(hipStreamIsCapturing(b.s, &st));
    MID_REQUIRE(st != hipStreamCaptureStatusNone, "record_end: the stream is not recording (no mid_record_begin on it)");
    hipGraph_t graph = nullptr;
    // (a call that failed inside the recording has invalidated it: EndCapture then reports the failure and the stream is usable again)
    hipError_t e = hipStreamEndCapture(b.s, &graph);
    if (e != hipSuccess || !graph) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        return set_error(MID_ERR_HIP, "record_end: the recording is invalid (%s): a call inside it failed or cannot be recorded",
                         e != hipSuccess ? hipGetErrorString(e) : "no graph");
    }
    mid_recording *r = new (std::nothrow) mid_recording{ctx, graph, nullptr, 0, 0};
    if (!r) { (void)hipGraphDestroy(graph); return set_error(MID_ERR_HIP, "record_end: out of host memory"); }
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) == hipSuccess && n > 0) {
        try {                                        // (no exception may cross the C ABI; the node counts are informational)
            std::vector<hipGraphNode_t> nodes(n);
            if (hipGraphGetNodes(graph, nodes.data(), &n) == hipSuccess) {
                r->n_nodes = (int)n;
                for (size_t i = 0; i < n; ++i) {
                    hipGraphNodeType t;
                    if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++r->n_kernels;
                }
            }
        } catch (...) { r->n_nodes = r->n_kernels = 0; }
    }
    (void)hipGetLastError();
    e = hipGraphInstantiate(&r->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        delete r;
        return set_error(MID_ERR_HIP, "record_end: hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    *out = r;
    return MID_OK;
}

extern "C" int mid_recording_submit(mid_recording *rec, void *stream)
{
    MID_REQUIRE(rec != nullptr, "recording_submit: NULL recording");
    Bind b(rec->ctx, stream);
    if (b.rc) return b.rc;
    MID_HIP(hipGraphLaunch(rec->exec, b.s));
    return MID_OK;
}

extern "C" int mid_recording_info(mid_recording *rec, int *n_nodes, int *n_kernels)
{
    MID_REQUIRE(rec != nullptr, "recording_info: NULL recording");
    if (n_nodes) *n_nodes = rec->n_nodes;
    if (n_kernels) *n_kernels = rec->n_kernels;
    return MID_OK;
}

extern "C" int mid_recording_destroy(mid_recording *rec)
{
    if (!rec) return MID_OK;
    (void)hipSetDevice(rec->ctx->device);
    if (rec->exec) (void)hipGraphExecDestroy(rec->exec);
    if (rec->graph) (void)hipGraphDestroy(rec->graph);
    delete rec;
    return MID_OK;
}
