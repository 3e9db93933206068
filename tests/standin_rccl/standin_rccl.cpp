// standin_rccl.cpp -- TEST-ONLY stand-in for librccl, loaded through MID_RCCL_LIBRARY by tests/test_gpu_sharded_multirank.py.
//
// Real RCCL refuses two ranks on one device, and the builder's pool hands out one-GPU boxes, so the C++ halo path
// (csrc/sharded.cpp: mid_comm_create_all, mid_nlm_temporal_sharded with world > 1, the CLI's --halo rccl threads) could not be
// EXECUTED with more than one rank.  This library implements just the ten entry points sharded.cpp binds, for ranks that are
// THREADS OF ONE PROCESS, possibly all on the same device: a send/receive pair becomes a device-to-device hipMemcpyAsync on the
// receiver's stream, ordered by events exactly as a transport would order it --
//   * the copy waits for everything the SENDER had queued on its stream when it posted the send,
//   * the RECEIVER's stream continues after the copy,
//   * the SENDER's stream continues only after the copy has read its buffer.
// Sends and receives of a pair of ranks are matched in issue order, like NCCL's.  ncclGroupEnd blocks the calling thread until
// the peers have posted their side, so every rank must call in from its own thread (as the CLI and the test do).
// What it is NOT: a transport.  It proves the library's plans, buffer bookkeeping, stream/event plumbing and failure handling with
// N > 1 ranks on real kernels; xGMI, RCCL's own kernels and multi-process rendezvous stay unexercised until a multi-GPU run.
// STANDIN_RCCL_FAIL_SEND_RANK=<r>: ncclSend fails on rank r (failure-path tests).
// STANDIN_RCCL_DELAY_MS=<ms>: every receive is held back by a one-wave kernel that spins for about that long on the receiver's
// exchange stream before the copy -- a slow wire, so that a test can see on the device timeline that interior launches do not
// wait for the halo and boundary launches do (the kernel reads the 100 MHz wall clock and always terminates).
// STANDIN_RCCL_COPY_WGS=<n>: the copy is done by a KERNEL of n workgroups x 256 threads instead of hipMemcpyAsync -- like RCCL's
// own send/receive, which are kernels that have to find compute units beside whatever else the device runs
// (tools/halo_overlap_probe.py uses it to see whether a busy device delays the exchange).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

__global__ void standin_hold(long long ticks)      // wall_clock64: constant 100 MHz on gfx950
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

__global__ void standin_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int copy_workgroups()
{
    const char *e = getenv("STANDIN_RCCL_COPY_WGS");
    const int n = e && *e ? atoi(e) : 0;
    return n > 0 && n <= 4096 ? n : 0;
}

long long delay_ticks()
{
    const char *e = getenv("STANDIN_RCCL_DELAY_MS");
    const double ms = e && *e ? atof(e) : 0.0;
    return ms > 0 && ms <= 2000 ? (long long)(ms * 1e5) : 0;          // capped at 2 s: a test knob, not a hang
}

struct Posted {                 // one posted send, waiting for (or matched with) its receive
    const void *src;
    size_t bytes;
    hipEvent_t ready;           // recorded on the sender's stream when the send was posted
    hipEvent_t done = nullptr;  // recorded by the receiver behind the copy
    bool matched = false;
    int device;
};

struct World {
    std::mutex mu;
    std::condition_variable cv;
    int n = 0, alive = 0;
    bool aborted = false;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Posted>>> wire;   // (src rank, dst rank) -> sends in issue order
};

struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
    std::shared_ptr<World> world;
    int rank = 0, device = 0;
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;      // unique id -> world (ncclCommInitRank)
unsigned long long g_next_id = 1;

thread_local int t_depth = 0;
thread_local std::vector<std::pair<Comm *, Op>> t_ops;

int fail_send_rank()
{
    const char *e = getenv("STANDIN_RCCL_FAIL_SEND_RANK");
    return e && *e ? atoi(e) : -1;
}

ncclResult_t flush()
{
    // 1. publish every send of this group
    std::vector<std::pair<Comm *, std::shared_ptr<Posted>>> mine;
    for (auto &co : t_ops) {
        Comm *c = co.first;
        const Op &op = co.second;
        if (!op.send) continue;
        auto p = std::make_shared<Posted>();
        p->src = op.buf; p->bytes = op.bytes; p->device = c->device;
        if (hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(p->ready, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> l(c->world->mu);
            c->world->wire[{c->rank, op.peer}].push_back(p);
        }
        c->world->cv.notify_all();
        mine.push_back({c, p});
    }
    // 2. receives: wait for the matching send, copy behind it on this rank's stream
    for (auto &co : t_ops) {
        Comm *c = co.first;
        const Op &op = co.second;
        if (op.send) continue;
        std::shared_ptr<Posted> p;
        {
            std::unique_lock<std::mutex> l(c->world->mu);
            auto &q = c->world->wire[{op.peer, c->rank}];
            c->world->cv.wait(l, [&] {
                if (c->world->aborted) return true;
                for (auto &s : q) if (!s->matched) return true;
                return false;
            });
            if (c->world->aborted) return ncclInternalError;
            for (auto &s : q) if (!s->matched) { p = s; break; }
            p->matched = true;
        }
        if (p->bytes != op.bytes) return ncclInvalidArgument;
        if (hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamWaitEvent(op.stream, p->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
        if (const long long ticks = delay_ticks()) {
            standin_hold<<<1, 64, 0, op.stream>>>(ticks);
            if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
        }
        if (const int wgs = copy_workgroups(); wgs && op.bytes % 16 == 0 && ((uintptr_t)op.buf | (uintptr_t)p->src) % 16 == 0) {
            standin_copy<<<wgs, 256, 0, op.stream>>>((const uint4 *)p->src, (uint4 *)op.buf, op.bytes / 16);
            if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
        } else if (hipMemcpyAsync(op.buf, p->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        hipEvent_t done;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(done, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> l(c->world->mu);
            p->done = done;
        }
        c->world->cv.notify_all();
    }
    // 3. sends complete (for the sender's stream) when the receiver's copy has read the buffer
    for (auto &cp : mine) {
        Comm *c = cp.first;
        std::shared_ptr<Posted> p = cp.second;
        {
            std::unique_lock<std::mutex> l(c->world->mu);
            c->world->cv.wait(l, [&] { return c->world->aborted || p->done != nullptr; });
            if (c->world->aborted) return ncclInternalError;
        }
        hipStream_t s = nullptr;
        for (auto &co : t_ops) if (co.first == c && co.second.send && co.second.buf == p->src) { s = co.second.stream; break; }
        if (hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamWaitEvent(s, p->done, 0) != hipSuccess) return ncclUnhandledCudaError;
        // (events are leaked on purpose: a test process lives for seconds, and destroying them here would need another rendezvous)
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    std::lock_guard<std::mutex> l(g_mu);
    snprintf(id->internal, sizeof id->internal, "standin-rccl-%llu", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto &slot = g_worlds[std::string(id.internal, strnlen(id.internal, sizeof id.internal))];
        if (!slot) { slot = std::make_shared<World>(); slot->n = nranks; }
        w = slot;
    }
    Comm *c = new Comm();
    c->world = w; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    { std::lock_guard<std::mutex> l(w->mu); ++w->alive; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devs)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    auto w = std::make_shared<World>();
    w->n = ndev; w->alive = ndev;
    for (int i = 0; i < ndev; ++i) {
        Comm *c = new Comm();
        c->world = w; c->rank = i; c->device = devs ? devs[i] : i;      // (the same device may appear more than once: that is the point)
        comms[i] = reinterpret_cast<ncclComm_t>(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    { std::lock_guard<std::mutex> l(c->world->mu); --c->world->alive; }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    { std::lock_guard<std::mutex> l(c->world->mu); c->world->aborted = true; --c->world->alive; }
    c->world->cv.notify_all();                    // ranks blocked in ncclGroupEnd return with an error
    // (the handle is leaked on purpose: the rank's own thread may be inside ncclGroupEnd with it right now)
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    ncclResult_t r = flush();
    t_ops.clear();
    return r;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !buf || peer < 0 || peer >= c->world->n || (dt != ncclUint8 && dt != ncclInt8)) return ncclInvalidArgument;
    if (c->rank == fail_send_rank()) return ncclSystemError;
    t_ops.push_back({c, Op{true, const_cast<void *>(buf), count, peer, s}});
    if (t_depth == 0) { ncclResult_t r = flush(); t_ops.clear(); return r; }
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !buf || peer < 0 || peer >= c->world->n || (dt != ncclUint8 && dt != ncclInt8)) return ncclInvalidArgument;
    t_ops.push_back({c, Op{false, buf, count, peer, s}});
    if (t_depth == 0) { ncclResult_t r = flush(); t_ops.clear(); return r; }
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "standin: no error";
    case ncclUnhandledCudaError: return "standin: a HIP call failed";
    case ncclSystemError: return "standin: injected system error";
    case ncclInternalError: return "standin: communicator aborted";
    case ncclInvalidArgument: return "standin: invalid argument";
    case ncclInvalidUsage: return "standin: invalid usage";
    default: return "standin: error";
    }
}

}  // extern "C"
