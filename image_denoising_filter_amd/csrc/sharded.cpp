// sharded.cpp -- frame-block sharding of an animation over GPUs, with the temporal-NLM halo exchanged GPU to GPU over
// RCCL (xGMI), from C++ (BASELINE configs[4]).
//
// The reference is single-device (deviceId{0}, src/main.cpp:1321) and filters ONE target per run against its
// neighbour frames (loop src/main.cpp:1577-1606); here every frame t of an n-frame sequence is an output, accumulated
// over the explicit window t-k..t+k (clipped at the sequence ends).  One rank per GPU owns a contiguous block of frames,
// resident in its HBM.  The only data another rank holds that a rank needs are the k frames on either side of its block:
// ONE exchange step -- ncclSend/ncclRecv inside one group, point to point with the neighbouring rank(s); no all-reduce,
// no all-gather -- after which the rank filters its block with the same kernels as a single GPU does.
//
//   stream (caller's)   : [e0] interior outputs ............................ [wait b1[0], b1[1]] done
//   exchange stream     : [wait e0] group{recv halo, send edges} [e1]            (highest priority)
//   boundary stream 0   : [wait e0] [wait e1] low-edge outputs ..... [b1[0]]     (lowest priority)
//   boundary stream 1   : [wait e0] [wait e1] high-edge outputs .... [b1[1]]     (lowest priority)
//
// Interior outputs (windows inside the rank's own block) are launched while the halo is in flight; the <= 2k boundary
// outputs wait for it ON STREAMS OF THEIR OWN, one per edge, so that their workgroups fill the tail of the interior launch
// instead of queueing behind it: at 8 frames per rank (64 frames on 8 GPUs, k = 2) the three launches are 9.03 + 4.5 + 4.5
// rounds of workgroups -- 19.9-20.0 ms back to back on one stream, 18.8 ms with each boundary launch on its own
// lowest-priority stream, which is what ONE launch over the 8 outputs takes (tools/boundary_stream_probe.py,
// profiles/r06_boundary_stream_priority.txt, LABNOTES R6.3).  Why the LOWEST priority: the runtime gives every priority
// level its own pool of four hardware queues, and a queue runs its packets in order.  At the default priority a boundary
// stream lands in the pool the caller's streams live in -- in a PyTorch process that pool is shared by torch's 32 pooled
// streams, and a boundary stream that shares the caller's queue gains nothing (19.95 ms; 18.8 only with
// GPU_MAX_HW_QUEUES=8 in the environment).  The lowest level is the library's alone: two boundary streams, two queues,
// whatever the application created; and workgroups of that level are dispatched where no interior workgroup waits, which
// is exactly the tail.  (ONE lowest-priority stream for both edges: 19.2 ms -- the second edge launch starts when the first
// has drained and pays a tail of its own.)  The caller's stream joins both at the end.  The launch plan (which outputs are interior, which frame table each launch sees) is the one
// image_denoising_filter_amd/sharding.py::block_launch_plan states and the gloo tests pin; mid_shard_* expose it as pure
// host functions so that the C++ and Python statements are tested against each other on the CPU.
//
// RCCL is bound at RUN time (dlopen of librccl.so.1 on the first mid_comm_* call): the library itself has no link-time
// dependency on it, single-GPU users never load it, and inside a PyTorch process the dlopen resolves to the copy torch
// has already mapped (same SONAME), so there is one RCCL per process.  MID_RCCL_LIBRARY=<path or soname>, read once on
// that first call, names the library to load instead of the default search (a site's own RCCL build); a library that
// cannot be loaded makes every mid_comm_* call return MID_ERR_UNSUPPORTED.
//
// mid_nlm_temporal_sharded is a COLLECTIVE: every rank must enter the exchange or none.  Everything that can fail
// locally -- argument checks, the stream rule, receive-buffer allocation -- therefore happens before the first RCCL
// call, and a rank that fails there has not touched the wire; its peers are then waiting for it, and the caller must
// abort (mid_comm_abort) or destroy the communicator on every rank.
#include "common.hpp"
#include <atomic>
#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string>
#include <vector>

using namespace mid;

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (reporting only; the test stand-in does not have them)
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char *forced = getenv("MID_RCCL_LIBRARY");
        std::vector<const char *> names;
        if (forced && *forced) names = {forced};
        else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *name : names) {
            x.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.so) break;
            const char *e = dlerror();          // (one call: dlerror() clears the message it returns)
            x.why += std::string(x.why.empty() ? "" : "; ") + "cannot load " + name + ": " + (e ? e : "?");
        }
        if (!x.so) return x;
        x.why.clear();
        auto sym = [&](const char *n) { void *p = dlsym(x.so, n); if (!p && x.why.empty()) x.why = std::string("librccl lacks ") + n; return p; };
        x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
        x.CommInitAll = (decltype(x.CommInitAll))sym("ncclCommInitAll");
        x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
        x.CommAbort = (decltype(x.CommAbort))sym("ncclCommAbort");
        x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
        x.Send = (decltype(x.Send))sym("ncclSend");
        x.Recv = (decltype(x.Recv))sym("ncclRecv");
        x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
        x.CommCount = (decltype(x.CommCount))dlsym(x.so, "ncclCommCount");
        x.CommUserRank = (decltype(x.CommUserRank))dlsym(x.so, "ncclCommUserRank");
        x.GetVersion = (decltype(x.GetVersion))dlsym(x.so, "ncclGetVersion");
        return x;
    }();
    return r;
}

#define MID_NCCL(call)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r__ = (call);                                                                         \
        if (r__ != ncclSuccess)                                                                            \
            return set_error(MID_ERR_HIP, "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

int need_rccl()
{
    Rccl &r = rccl();
    if (!r.so || !r.why.empty()) return set_error(MID_ERR_UNSUPPORTED, "RCCL is not available: %s", r.why.c_str());
    return MID_OK;
}

// ---- pure host logic (also exported: mid_shard_block / mid_shard_halo_plan / mid_shard_launch_plan) ----
void block_of(int n, int world, int rank, int &start, int &count)
{
    const int q = n / world, r = n % world;                 // the first n % world ranks get one extra frame
    start = rank * q + (rank < r ? rank : r);
    count = q + (rank < r ? 1 : 0);
}

int owner_of(int n, int world, int f)
{
    const int q = n / world, r = n % world;
    const int big = r * (q + 1);
    if (f < big) return f / (q + 1);
    return q ? r + (f - big) / q : world - 1;
}

// frames rank `r` needs but does not own, ascending
void needs_of(int n, int world, int k, int r, std::vector<int> &out)
{
    out.clear();
    int s, c;
    block_of(n, world, r, s, c);
    if (c == 0 || k == 0) return;
    const int lo = s - k < 0 ? 0 : s - k, hi = s + c - 1 + k > n - 1 ? n - 1 : s + c - 1 + k;
    for (int f = lo; f <= hi; ++f)
        if (f < s || f >= s + c) out.push_back(f);
}

struct Xfer { int peer, frame; };

// recv: this rank's needs, ascending frame (the peer is the owner).  send: for every OTHER rank in ascending rank order,
// the frames it needs that this rank owns, ascending.  Between any two ranks a and b the frames a sends to b are, in
// order, exactly the frames b receives from a: both sides enumerate "needs(b) owned by a" ascending -- RCCL matches the
// sends and receives of a pair in issue order.
void halo_plan(int n, int world, int k, int rank, std::vector<Xfer> &recv, std::vector<Xfer> &send)
{
    recv.clear(); send.clear();
    std::vector<int> nd;
    needs_of(n, world, k, rank, nd);
    for (int f : nd) recv.push_back({owner_of(n, world, f), f});
    int s, c;
    block_of(n, world, rank, s, c);
    if (c == 0) return;
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        needs_of(n, world, k, r, nd);
        for (int f : nd)
            if (f >= s && f < s + c) send.push_back({r, f});
    }
}

struct Launch { int interior, w_lo, w_hi, first, count, off; };

// sharding.py::block_launch_plan, statement for statement
void launch_plan(int n, int world, int k, int rank, std::vector<Launch> &plan)
{
    plan.clear();
    int start, count;
    block_of(n, world, rank, start, count);
    if (count == 0) return;
    const int lo_int = start == 0 ? start : start + k;
    const int hi_int = start + count == n ? start + count : start + count - k;   // exclusive
    if (hi_int > lo_int) {
        const int w_lo = start > lo_int - k ? start : lo_int - k;
        const int w_hi = start + count - 1 < hi_int - 1 + k ? start + count - 1 : hi_int - 1 + k;
        plan.push_back({1, w_lo, w_hi, lo_int - w_lo, hi_int - lo_int, lo_int - start});
    }
    const int edges[2][2] = {{start, lo_int < start + count ? lo_int : start + count},
                             {hi_int > lo_int ? hi_int : lo_int, start + count}};
    for (auto &e : edges) {
        const int a = e[0], b = e[1];
        if (b > a) {
            const int w_lo = a - k < 0 ? 0 : a - k, w_hi = b - 1 + k > n - 1 ? n - 1 : b - 1 + k;
            plan.push_back({0, w_lo, w_hi, a - w_lo, b - a, a - start});
        }
    }
}

}  // namespace

struct mid_comm {
    mid_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t xs = nullptr;               // exchange stream (highest priority the device offers: see comm_finish_create)
    int xs_priority = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr, x0 = nullptr;
    hipEvent_t l0 = nullptr, l1 = nullptr, lq = nullptr;   // mid_comm_loopback's own events: queued (caller's stream), start and end (exchange stream)
    bool have_loop = false;
    hipStream_t bs[2] = {nullptr, nullptr}; // boundary streams (lowest priority), one per edge of the block: the launches that wait for the halo run here, in the interior launches' tail
    hipEvent_t b1[2] = {nullptr, nullptr};  // end of the boundary launch on bs[j]; the caller's stream waits for them before `done`
    bool bs_used[2] = {false, false};       // this call queued something on bs[j] (the join is then owed, also on an error path)
    hipEvent_t i1 = nullptr;                // end of the interior launches on the caller's stream (timeline only)
    hipEvent_t done = nullptr;              // end of the last sharded call's launches, on the stream it was issued on
    bool have_i1 = false, have_done = false;
    std::string issued;                     // host-side issue order of the last call: X I.. W B.. (mid_comm_last_issue_order)
    hipStream_t last_stream = nullptr;      // that stream (valid while has_last)
    bool has_last = false;
    std::vector<void *> halo;               // device buffers for received frames, grown on demand
    size_t halo_bytes = 0;                  // size of each
    std::vector<void *> retired;            // buffers of an earlier, smaller frame size: freed once `done` has passed
    size_t last_recv = 0, last_sent = 0;
    bool timed = false;
    std::atomic<bool> aborted{false};       // set by mid_comm_abort, possibly from another thread than the one inside a call
};

// The exchange stream gets the HIGHEST priority the device offers.  RCCL's send/receive are kernels (a few workgroups
// per channel) that must find compute units beside an interior NLM launch of thousands of workgroups already queued on
// the caller's stream; at equal priority the hardware scheduler may serve them only as interior workgroups retire, and
// the halo -- which the boundary launches wait for -- would arrive late for no reason.  Priority only orders dispatch
// of WAITING workgroups, it pre-empts nothing, so the interior launch loses at most the few CUs the transport needs.
// Whether this hides the exchange completely is what mid_comm_last_timeline reports on the first multi-GPU run.
static int comm_finish_create(mid_comm *c)
{
    int least = 0, greatest = 0;
    MID_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));         // numerically lower = higher priority
    c->xs_priority = greatest;
    MID_HIP(hipStreamCreateWithPriority(&c->xs, hipStreamNonBlocking, greatest));
    MID_HIP(hipEventCreate(&c->e0));
    MID_HIP(hipEventCreate(&c->e1));
    MID_HIP(hipEventCreate(&c->x0));
    MID_HIP(hipEventCreate(&c->l0));
    MID_HIP(hipEventCreate(&c->l1));
    MID_HIP(hipEventCreate(&c->lq));
    MID_HIP(hipEventCreate(&c->i1));
    MID_HIP(hipEventCreate(&c->done));
    for (int j = 0; j < 2; ++j) {
        MID_HIP(hipStreamCreateWithPriority(&c->bs[j], hipStreamNonBlocking, least));
        MID_HIP(hipEventCreateWithFlags(&c->b1[j], hipEventDisableTiming));
    }
    return MID_OK;
}

// Receive buffers: `n` of `bytes` each.  A larger frame size retires the old buffers instead of freeing them -- launches
// of the previous call may still read them, and both hipFree and a device-wide synchronisation would stall the caller's
// streams mid-call; they are released once that call's `done` event has passed (or with the communicator).
static int reserve_halo(mid_comm *c, size_t bytes, size_t n)
{
    if (!c->retired.empty() && (!c->has_last || hipEventQuery(c->done) == hipSuccess)) {
        for (void *q : c->retired) (void)hipFree(q);
        c->retired.clear();
    }
    if (bytes > c->halo_bytes) {
        c->retired.insert(c->retired.end(), c->halo.begin(), c->halo.end());
        c->halo.clear();
        c->halo_bytes = bytes;
    }
    while (c->halo.size() < n) { void *q = nullptr; MID_HIP(hipMalloc(&q, c->halo_bytes)); c->halo.push_back(q); }
    return MID_OK;
}

extern "C" int mid_shard_block(int n_frames, int world, int rank, int *start, int *count)
{
    MID_REQUIRE(n_frames >= 0 && world >= 1 && rank >= 0 && rank < world && start && count, "shard_block: bad argument");
    block_of(n_frames, world, rank, *start, *count);
    return MID_OK;
}

extern "C" int mid_shard_halo_plan(int n_frames, int world, int k, int rank, int cap,
                                   int *recv_peer, int *recv_frame, int *n_recv,
                                   int *send_peer, int *send_frame, int *n_send)
{
    MID_REQUIRE(n_frames >= 0 && world >= 1 && rank >= 0 && rank < world && k >= 0 && n_recv && n_send, "shard_halo_plan: bad argument");
    std::vector<Xfer> rv, sd;
    if (world > 1) halo_plan(n_frames, world, k, rank, rv, sd);
    *n_recv = (int)rv.size(); *n_send = (int)sd.size();
    MID_REQUIRE((int)rv.size() <= cap && (int)sd.size() <= cap, "shard_halo_plan: %zu receives / %zu sends do not fit cap %d", rv.size(), sd.size(), cap);
    for (size_t i = 0; i < rv.size(); ++i) { if (recv_peer) recv_peer[i] = rv[i].peer; if (recv_frame) recv_frame[i] = rv[i].frame; }
    for (size_t i = 0; i < sd.size(); ++i) { if (send_peer) send_peer[i] = sd[i].peer; if (send_frame) send_frame[i] = sd[i].frame; }
    return MID_OK;
}

extern "C" int mid_shard_launch_plan(int n_frames, int world, int k, int rank, int cap, int *rows /* cap x 6 */, int *n_rows)
{
    MID_REQUIRE(n_frames >= 0 && world >= 1 && rank >= 0 && rank < world && k >= 0 && rows && n_rows, "shard_launch_plan: bad argument");
    std::vector<Launch> pl;
    launch_plan(n_frames, world, k, rank, pl);
    *n_rows = (int)pl.size();
    MID_REQUIRE((int)pl.size() <= cap, "shard_launch_plan: %zu rows do not fit cap %d", pl.size(), cap);
    for (size_t i = 0; i < pl.size(); ++i) {
        int *r = rows + 6 * i;
        r[0] = pl[i].interior; r[1] = pl[i].w_lo; r[2] = pl[i].w_hi; r[3] = pl[i].first; r[4] = pl[i].count; r[5] = pl[i].off;
    }
    return MID_OK;
}

extern "C" int mid_comm_unique_id(uint8_t id[MID_COMM_ID_BYTES])
{
    static_assert(MID_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    MID_REQUIRE(id != nullptr, "comm_unique_id: id is NULL");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId u;
    MID_NCCL(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, MID_COMM_ID_BYTES);
    return MID_OK;
}

extern "C" int mid_comm_create(mid_ctx *ctx, const uint8_t id[MID_COMM_ID_BYTES], int rank, int world, mid_comm **out)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    MID_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, "comm_create: bad argument (rank %d of %d)", rank, world);
    *out = nullptr;
    if (int rc = need_rccl()) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, MID_COMM_ID_BYTES);
    mid_comm *c = new mid_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return set_error(MID_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl().GetErrorString(r)); }
    if (int rc = comm_finish_create(c)) { (void)mid_comm_destroy(c); return rc; }
    *out = c;
    return MID_OK;
}

extern "C" int mid_comm_create_all(mid_ctx *const *ctxs, int world, mid_comm **out)
{
    MID_REQUIRE(ctxs && out && world >= 1 && world <= 64, "comm_create_all: bad argument");
    for (int i = 0; i < world; ++i) { MID_REQUIRE(ctxs[i], "comm_create_all: context %d is NULL", i); out[i] = nullptr; }
    if (int rc = need_rccl()) return rc;
    std::vector<int> devs(world);
    std::vector<ncclComm_t> comms(world, nullptr);
    for (int i = 0; i < world; ++i) devs[i] = ctxs[i]->device;
    MID_NCCL(rccl().CommInitAll(comms.data(), world, devs.data()));
    for (int i = 0; i < world; ++i) {
        mid_comm *c = new mid_comm();
        c->ctx = ctxs[i]; c->rank = i; c->world = world; c->comm = comms[i];
        out[i] = c;
    }
    for (int i = 0; i < world; ++i) {
        Bind b(ctxs[i], nullptr);
        int rc = b.rc ? b.rc : comm_finish_create(out[i]);
        if (rc) { for (int j = 0; j < world; ++j) { (void)mid_comm_destroy(out[j]); out[j] = nullptr; } return rc; }
    }
    return MID_OK;
}

extern "C" int mid_comm_destroy(mid_comm *c)
{
    if (!c) return MID_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->has_last && c->done) (void)hipEventSynchronize(c->done);      // launches that still read the receive buffers
    for (void *p : c->halo) (void)hipFree(p);
    for (void *p : c->retired) (void)hipFree(p);
    if (c->comm && !c->aborted) (void)rccl().CommDestroy(c->comm);
    if (c->e0) (void)hipEventDestroy(c->e0);
    if (c->e1) (void)hipEventDestroy(c->e1);
    if (c->x0) (void)hipEventDestroy(c->x0);
    for (hipEvent_t e : {c->l0, c->l1, c->lq}) if (e) (void)hipEventDestroy(e);
    if (c->i1) (void)hipEventDestroy(c->i1);
    for (int j = 0; j < 2; ++j) {
        if (c->b1[j]) (void)hipEventDestroy(c->b1[j]);
        if (c->bs[j]) { (void)hipStreamSynchronize(c->bs[j]); (void)hipStreamDestroy(c->bs[j]); }
    }
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    delete c;
    return MID_OK;
}

// ncclCommAbort: tears this rank's connections down WITHOUT waiting for outstanding operations -- the way out when a rank
// has failed before (or inside) a collective call and its peers would otherwise wait for it for ever.  The handle stays
// valid for mid_comm_destroy only; every other call on it returns MID_ERR_INVALID.
extern "C" int mid_comm_abort(mid_comm *c)
{
    MID_REQUIRE(c != nullptr, "comm_abort: comm is NULL");
    if (c->aborted.exchange(true)) return MID_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->comm) MID_NCCL(rccl().CommAbort(c->comm));      // (frees the communicator: mid_comm_destroy skips ncclCommDestroy)
    return MID_OK;
}

extern "C" int mid_comm_reserve(mid_comm *c, size_t max_frame_bytes, int k)
{
    MID_REQUIRE(c && !c->aborted && max_frame_bytes > 0 && k >= 0, "comm_reserve: bad argument");
    Bind b(c->ctx, nullptr);
    if (b.rc) return b.rc;
    return reserve_halo(c, max_frame_bytes, c->world > 1 ? 2 * (size_t)k : 0);
}

extern "C" int mid_comm_rank(mid_comm *c, int *rank, int *world)
{
    MID_REQUIRE(c != nullptr, "comm_rank: comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return MID_OK;
}

// ncclSend + ncclRecv addressed to this very rank inside one group on the exchange stream: the halo exchange's call
// pattern minus the wire -- what can be exercised on a one-GPU box (tests, smoke); also a cheap liveness probe of a comm.
// It records its OWN events (lq on the caller's stream, l0/l1 around the group on the exchange stream): the timeline of the
// last sharded call (e0/x0/e1/i1/done) stays that call's, whatever is looped back afterwards.
extern "C" int mid_comm_loopback(mid_comm *c, const void *src, void *dst, size_t bytes, void *stream)
{
    MID_REQUIRE(c && src && dst && bytes > 0, "comm_loopback: bad argument");
    MID_REQUIRE(!c->aborted, "comm_loopback: the communicator was aborted");
    Bind b(c->ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(b.s, "mid_comm_loopback (RCCL on the exchange stream)")) return rc;
    c->have_loop = false;
    MID_HIP(hipEventRecord(c->lq, b.s));
    MID_HIP(hipStreamWaitEvent(c->xs, c->lq, 0));
    MID_HIP(hipEventRecord(c->l0, c->xs));
    MID_NCCL(rccl().GroupStart());
    ncclResult_t r1 = rccl().Recv(dst, bytes, ncclUint8, c->rank, c->comm, c->xs);
    ncclResult_t r2 = rccl().Send(src, bytes, ncclUint8, c->rank, c->comm, c->xs);
    ncclResult_t r3 = rccl().GroupEnd();
    if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess)
        return set_error(MID_ERR_HIP, "comm_loopback: ncclRecv/ncclSend/ncclGroupEnd: %s / %s / %s", rccl().GetErrorString(r1), rccl().GetErrorString(r2), rccl().GetErrorString(r3));
    MID_HIP(hipEventRecord(c->l1, c->xs));
    MID_HIP(hipStreamWaitEvent(b.s, c->l1, 0));
    c->have_loop = true;
    return MID_OK;
}

// (waits for it) the last loopback on the device's clock, ms from the moment the caller's stream reached the call:
// t[0] start, t[1] end of the send+receive group on the exchange stream.
extern "C" int mid_comm_last_loopback(mid_comm *c, float t_ms[2])
{
    MID_REQUIRE(c && t_ms, "comm_last_loopback: NULL argument");
    t_ms[0] = t_ms[1] = 0.f;
    MID_REQUIRE(c->have_loop, "comm_last_loopback: no mid_comm_loopback has been queued on this communicator");
    Bind b(c->ctx, nullptr);
    if (b.rc) return b.rc;
    MID_HIP(hipEventSynchronize(c->l1));
    MID_HIP(hipEventElapsedTime(&t_ms[0], c->lq, c->l0));
    MID_HIP(hipEventElapsedTime(&t_ms[1], c->lq, c->l1));
    return MID_OK;
}

extern "C" int mid_nlm_temporal_sharded(mid_comm *c, const mid_nlm_params *p, const void *const *block, int n_frames, int k,
                                        mid_pixel *const *out, void *stream)
{
    MID_REQUIRE(c && p && n_frames >= 1 && k >= 0, "nlm_temporal_sharded: bad argument");
    MID_REQUIRE(!c->aborted, "nlm_temporal_sharded: the communicator was aborted");
    Bind b(c->ctx, stream);
    if (b.rc) return b.rc;
    if (int rc = refuse_if_recording(b.s, "mid_nlm_temporal_sharded (RCCL, exchange and boundary streams)")) return rc;
    // Stream rule.  The receive buffers, e0/e1 and the exchange stream are reused from call to call, ordered behind the
    // previous call's launches only through the stream both calls are issued on.  A call on ANOTHER stream is accepted
    // only once the previous call has finished (the caller synchronised its stream, or the work simply is done);
    // otherwise it would race on the halo buffers silently.
    if (c->has_last && c->last_stream != b.s) {
        const hipError_t q = hipEventQuery(c->done);
        MID_REQUIRE(q != hipErrorNotReady,
                    "nlm_temporal_sharded: the previous call on this communicator was issued on another stream and is still in flight; "
                    "use one stream per communicator or synchronise the earlier stream first (mid_stream_sync)");
        MID_HIP(q);
    }
    int start, count;
    block_of(n_frames, c->world, c->rank, start, count);
    MID_REQUIRE(count == 0 || (block && out), "nlm_temporal_sharded: NULL table");
    MID_REQUIRE(p->width > 0 && p->height > 0 && (p->format == MID_FMT_RGBA32F || p->format == MID_FMT_RGBA8), "nlm_temporal_sharded: bad params");
    for (int i = 0; i < count; ++i) MID_REQUIRE(block[i] && out[i], "nlm_temporal_sharded: frame %d of the block is NULL", i);
    const size_t frame_bytes = (size_t)p->width * p->height * (p->format == MID_FMT_RGBA8 ? 4 : 16);

    std::vector<Xfer> rv, sd;
    if (c->world > 1) halo_plan(n_frames, c->world, k, c->rank, rv, sd);
    c->last_recv = rv.size() * frame_bytes; c->last_sent = sd.size() * frame_bytes;
    c->timed = false; c->have_i1 = false; c->have_done = false; c->issued.clear();

    // receive buffers (kept across calls; a call's receives are ordered after the previous call's last readers through
    // e0 on the one stream the stream rule above enforces).  This is the last step that can fail for a LOCAL reason
    // (out of memory); after it only a HIP call on a broken device or RCCL itself can make the rank return early.
    if (int rc = reserve_halo(c, frame_bytes, rv.size())) return rc;

    // From the first thing queued on the caller's stream, however this call ends -- a launch that fails half way through
    // the plan, a HIP error after the group was closed -- `done` is recorded behind whatever WAS queued and
    // last_stream/has_last name this call: the stream rule of the next call and reserve_halo's "may the retired buffers
    // be freed" then test this call's work, not an older call's.
    struct Finish {
        mid_comm *c; hipStream_t s; bool armed = false;
        ~Finish()
        {
            if (!armed) return;
            for (int j = 0; j < 2; ++j) {
                if (!c->bs_used[j]) continue;                      // the caller's stream continues only after the boundary streams' work
                if (hipEventRecord(c->b1[j], c->bs[j]) != hipSuccess || hipStreamWaitEvent(s, c->b1[j], 0) != hipSuccess)
                    (void)hipStreamSynchronize(c->bs[j]);          // (a broken device: fall back to a host-side join)
                c->bs_used[j] = false;
            }
            // `done` is what the next call's stream rule and reserve_halo's "may the retired buffers go" test: it must stand
            // behind THIS call's work.  If it cannot be recorded (a broken device), a stale `done` of an older call would answer
            // for this one -- so wait the stream out on the host instead and leave no call to be asked about.
            c->have_done = hipEventRecord(c->done, s) == hipSuccess;
            c->last_stream = s; c->has_last = c->have_done;
            if (!c->have_done) (void)hipStreamSynchronize(s);
        }
    } finish{c, b.s};

    MID_HIP(hipEventRecord(c->e0, b.s));                           // t = 0 of the timeline; the block's frames (and the halo buffers' last readers) are done
    finish.armed = true;
    if (!rv.empty() || !sd.empty()) {
        MID_HIP(hipStreamWaitEvent(c->xs, c->e0, 0));
        MID_HIP(hipEventRecord(c->x0, c->xs));
        MID_NCCL(rccl().GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (size_t i = 0; i < rv.size() && bad == ncclSuccess; ++i) bad = rccl().Recv(c->halo[i], frame_bytes, ncclUint8, rv[i].peer, c->comm, c->xs);
        for (size_t i = 0; i < sd.size() && bad == ncclSuccess; ++i) bad = rccl().Send(block[sd[i].frame - start], frame_bytes, ncclUint8, sd[i].peer, c->comm, c->xs);
        ncclResult_t end = rccl().GroupEnd();                      // always closed, even after a failed call inside
        if (bad != ncclSuccess || end != ncclSuccess)
            return set_error(MID_ERR_HIP, "halo exchange: %s", rccl().GetErrorString(bad != ncclSuccess ? bad : end));
        MID_HIP(hipEventRecord(c->e1, c->xs));
        c->timed = true;
        c->issued += 'X';
    }
    if (count == 0) return MID_OK;

    std::vector<Launch> plan;
    launch_plan(n_frames, c->world, k, c->rank, plan);
    auto frame_ptr = [&](int f) -> const void * {
        if (f >= start && f < start + count) return block[f - start];
        for (size_t i = 0; i < rv.size(); ++i) if (rv[i].frame == f) return c->halo[i];
        return nullptr;
    };
    auto launch = [&](const Launch &L, hipStream_t ls) -> int {
        std::vector<const void *> tbl(L.w_hi - L.w_lo + 1);
        for (int f = L.w_lo; f <= L.w_hi; ++f) {
            tbl[f - L.w_lo] = frame_ptr(f);
            MID_REQUIRE(tbl[f - L.w_lo], "nlm_temporal_sharded: frame %d is neither in the block nor in the halo (plan error)", f);
        }
        return mid_nlm_temporal(c->ctx, p, tbl.data(), (int)tbl.size(), k, L.first, L.count, out + L.off, ls);
    };
    // interior launches first, on the caller's stream ...
    for (const Launch &L : plan) {
        if (!L.interior) continue;
        if (int rc = launch(L, b.s)) return rc;
        c->issued += 'I';
    }
    MID_HIP(hipEventRecord(c->i1, b.s));
    c->have_i1 = true;
    // ... then the boundary launches, each edge of the block on its own lowest-priority stream: ordered behind the block's
    // frames (e0) and -- only here, after every interior launch has been queued -- told to wait for the exchange
    int edge = 0;
    for (const Launch &L : plan) {
        if (L.interior) continue;
        const int j = edge++ & 1;
        if (!c->bs_used[j]) {
            c->bs_used[j] = true;
            MID_HIP(hipStreamWaitEvent(c->bs[j], c->e0, 0));
            if (c->timed) { MID_HIP(hipStreamWaitEvent(c->bs[j], c->e1, 0)); c->issued += 'W'; }
        }
        if (int rc = launch(L, c->bs[j])) return rc;
        c->issued += 'B';
    }
    return MID_OK;                                                 // (`finish` records `done`)
}

extern "C" int mid_comm_last_exchange(mid_comm *c, size_t *bytes_recv, size_t *bytes_sent, float *exchange_ms)
{
    MID_REQUIRE(c != nullptr, "comm_last_exchange: comm is NULL");
    if (bytes_recv) *bytes_recv = c->last_recv;
    if (bytes_sent) *bytes_sent = c->last_sent;
    if (exchange_ms) {
        *exchange_ms = 0.f;
        if (c->timed) {
            Bind b(c->ctx, nullptr);
            if (b.rc) return b.rc;
            MID_HIP(hipEventSynchronize(c->e1));
            MID_HIP(hipEventElapsedTime(exchange_ms, c->x0, c->e1));
        }
    }
    return MID_OK;
}

// Device timeline of the last sharded call, ms from its first event on the caller's stream: [0] exchange start, [1] exchange
// end (0, 0 when nothing was exchanged), [2] end of the interior launches, [3] end of the call's last launch.  The share of
// the exchange that ran while interior launches were still executing is (min(t1, t2) - t0) / (t1 - t0), clamped to [0, 1].
extern "C" int mid_comm_last_timeline(mid_comm *c, float t_ms[4])
{
    MID_REQUIRE(c && t_ms, "comm_last_timeline: NULL argument");
    t_ms[0] = t_ms[1] = t_ms[2] = t_ms[3] = 0.f;
    if (!c->has_last || !c->have_done) return MID_OK;
    Bind b(c->ctx, nullptr);
    if (b.rc) return b.rc;
    MID_HIP(hipEventSynchronize(c->done));
    if (c->timed) {
        MID_HIP(hipEventSynchronize(c->e1));
        MID_HIP(hipEventElapsedTime(&t_ms[0], c->e0, c->x0));
        MID_HIP(hipEventElapsedTime(&t_ms[1], c->e0, c->e1));
    }
    if (c->have_i1) MID_HIP(hipEventElapsedTime(&t_ms[2], c->e0, c->i1));
    MID_HIP(hipEventElapsedTime(&t_ms[3], c->e0, c->done));
    return MID_OK;
}

// What the last sharded call put on its streams, in host issue order: 'X' the exchange group (exchange stream), 'I' an
// interior launch (caller's stream), 'W' a boundary stream's wait for the exchange, 'B' a boundary launch (on that boundary
// stream): "X I.. W B [W B]".  The overlap of halo and interior compute is structural when every 'I' precedes the first 'W'
// (tests/test_gpu_sharded_multirank.py asserts it).
extern "C" int mid_comm_last_issue_order(mid_comm *c, char *buf, size_t buflen)
{
    MID_REQUIRE(c && buf && buflen > 0, "comm_last_issue_order: bad argument");
    snprintf(buf, buflen, "%s", c->issued.c_str());
    return MID_OK;
}

extern "C" int mid_comm_boundary_priority(mid_comm *c, int priority[2])
{
    MID_REQUIRE(c && priority, "comm_boundary_priority: NULL argument");
    Bind b(c->ctx, nullptr);
    if (b.rc) return b.rc;
    for (int j = 0; j < 2; ++j) MID_HIP(hipStreamGetPriority(c->bs[j], &priority[j]));
    return MID_OK;
}

extern "C" int mid_comm_stream_priority(mid_comm *c, int *priority, int *least, int *greatest)
{
    MID_REQUIRE(c != nullptr, "comm_stream_priority: comm is NULL");
    Bind b(c->ctx, nullptr);
    if (b.rc) return b.rc;
    int lo = 0, hi = 0, pr = 0;
    MID_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    MID_HIP(hipStreamGetPriority(c->xs, &pr));
    if (priority) *priority = pr;
    if (least) *least = lo;
    if (greatest) *greatest = hi;
    return MID_OK;
}

// What RCCL ITSELF says about this communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion) -- as opposed to
// mid_comm_rank, which returns what the caller passed to mid_comm_create.  -1 where the loaded library lacks the call.
extern "C" int mid_comm_rccl_info(mid_comm *c, int *nranks, int *user_rank, int *version)
{
    MID_REQUIRE(c != nullptr, "comm_rccl_info: comm is NULL");
    MID_REQUIRE(!c->aborted, "comm_rccl_info: the communicator was aborted");
    if (int rc = need_rccl()) return rc;
    int n = -1, r = -1, v = -1;
    if (rccl().CommCount) MID_NCCL(rccl().CommCount(c->comm, &n));
    if (rccl().CommUserRank) MID_NCCL(rccl().CommUserRank(c->comm, &r));
    if (rccl().GetVersion) MID_NCCL(rccl().GetVersion(&v));
    if (nranks) *nranks = n;
    if (user_rank) *user_rank = r;
    if (version) *version = v;
    return MID_OK;
}
