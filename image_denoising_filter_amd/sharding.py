"""Frame-level data parallelism for animation sequences (BASELINE config 5).

The reference is single-device (deviceId{0}, src/main.cpp:1321); this is new capability with the
reference's multi-frame semantics: output frame t accumulates over neighbour frames t-k..t+k
(src/main.cpp:1577-1606 with an explicit, clipped window).

One process per GPU.  Rank r owns a contiguous block of frames.  Bilateral / single-frame NLM need
nothing from other ranks (replicas).  Temporal NLM needs the k frames on either side of the block:
ONE exchange step -- each rank sends its first k frames to rank r-1 and its last k to rank r+1
(point-to-point over xGMI with backend "nccl" = RCCL; "gloo" in the CPU tests) -- and no other
collective.  Sequence ends are clipped, not wrapped.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def partition(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous (start, count) per rank; the first n_frames % world ranks get one extra frame."""
    if n_frames < 0 or world < 1:
        raise ValueError("bad partition request")
    q, r = divmod(n_frames, world)
    out, s = [], 0
    for i in range(world):
        c = q + (1 if i < r else 0)
        out.append((s, c))
        s += c
    return out


def halo_plan(n_frames: int, world: int, k: int, rank: int):
    """Which global frame indices this rank receives from / sends to which rank.
    Returns (recv, send): lists of (peer_rank, [global frame ids]).  Handles blocks shorter than k
    (a halo may then span several ranks) and empty blocks."""
    parts = partition(n_frames, world)
    owner = {}
    for r, (s, c) in enumerate(parts):
        for f in range(s, s + c):
            owner[f] = r

    def needs(r):
        s, c = parts[r]
        if c == 0:
            return []
        lo, hi = max(0, s - k), min(n_frames - 1, s + c - 1 + k)
        return [f for f in range(lo, hi + 1) if not (s <= f < s + c)]

    recv = {}
    for f in needs(rank):
        recv.setdefault(owner[f], []).append(f)
    send = {}
    for r in range(world):
        if r == rank:
            continue
        for f in needs(r):
            if owner[f] == rank:
                send.setdefault(r, []).append(f)
    return sorted(recv.items()), sorted(send.items())


def exchange_halo(local: Sequence[torch.Tensor], n_frames: int, k: int, group=None):
    """Returns {global frame id: tensor} for every frame this rank needs (its own block + halo).
    `local` are this rank's frames in order.  One batch of isend/irecv; no-op for world_size 1 or k 0."""
    have, reqs = start_halo_exchange(local, n_frames, k, group)
    for req in reqs:
        req.wait()
    if have.finalize:
        have.finalize()
    if reqs:
        _order_after_halo(local)
    return have


class _Have(dict):
    """{global frame id: tensor}; `finalize` (when set) must be called after the requests have been waited for -- it
    moves halo frames that were staged through host memory onto the device."""
    finalize = None


def _must_stage(tensor, group):
    """Device tensors cannot travel over a backend that only moves host memory (gloo): stage them through the host.
    The test is for a group whose backend IS "gloo".  A group created with a per-device backend map
    ("cpu:gloo,cuda:nccl") reports that whole string: its device tensors go over nccl (RCCL) directly, which is what we
    want, so it is deliberately not staged."""
    return bool(getattr(tensor, "is_cuda", False)) and dist.is_initialized() and dist.get_backend(group) == "gloo"


def start_halo_exchange(local: Sequence[torch.Tensor], n_frames: int, k: int, group=None):
    """Non-blocking form of exchange_halo: posts the isend/irecv batch and returns (have, requests).
    `have` already contains this rank's own frames; halo entries are valid only after every request
    in `requests` has been waited for (and `have.finalize()` has run, if it is set).
    With backend "nccl" (RCCL) device tensors go GPU to GPU over xGMI.  With "gloo" and device tensors the frames are
    staged through host memory on both sides (a rehearsal path: one GPU box, several ranks on one device)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, count = partition(n_frames, world)[rank]
    if len(local) != count:
        raise ValueError(f"rank {rank} owns {count} frames, got {len(local)}")
    have = _Have((start + i, t) for i, t in enumerate(local))
    if world == 1 or k == 0 or count == 0:
        return have, []
    recv, send = halo_plan(n_frames, world, k, rank)
    stage = _must_stage(local[0], group)
    ops, staged = [], []
    for peer, ids in recv:
        for f in ids:
            buf = torch.empty(local[0].shape, dtype=local[0].dtype, device="cpu") if stage else torch.empty_like(local[0])
            if stage:
                staged.append((f, buf))
            else:
                have[f] = buf
            ops.append(dist.P2POp(dist.irecv, buf, peer, group=group, tag=f))
    for peer, ids in send:
        for f in ids:
            ops.append(dist.P2POp(dist.isend, have[f].cpu() if stage else have[f], peer, group=group, tag=f))
    if stage:
        def finalize():
            for f, buf in staged:
                have[f] = buf.to(local[0].device)
            torch.cuda.current_stream(local[0].device).synchronize()     # the launches that follow may use any stream
        have.finalize = finalize
    return have, (dist.batch_isend_irecv(ops) if ops else [])


def _order_after_halo(tensors):
    """Stream rule for device backends.  With backend "nccl" (RCCL) `req.wait()` only makes the CURRENT torch
    stream wait for the transfer; kernels launched through the C-ABI run on the stream handle they are given,
    and the handle 0 (torch's default stream) means "the context's own stream" there -- a stream RCCL knows
    nothing about.  So: callers launch on torch.cuda.current_stream().cuda_stream with a non-default stream
    current (what bench.py does), and if the current stream IS the default one this helper falls back to a
    host-side synchronize of it, which orders every later launch on any stream after the received halo."""
    if tensors and getattr(tensors[0], "is_cuda", False):
        cur = torch.cuda.current_stream(tensors[0].device)
        if cur.cuda_stream == 0:
            cur.synchronize()


def launch_stream_for(tensor):
    """The stream handle a C-ABI launch must use to be ordered after RCCL halo receives on `tensor`'s device:
    the current torch stream.  Raises if that is the default stream (handle 0 = the library's own stream)."""
    if not getattr(tensor, "is_cuda", False):
        return None
    h = torch.cuda.current_stream(tensor.device).cuda_stream
    if h == 0:
        raise RuntimeError("make a non-default torch stream current (torch.cuda.set_stream) before launching on "
                           "halo frames: stream handle 0 means the library's own stream to the C-ABI, which is "
                           "not ordered after RCCL's receives")
    return h


def temporal_block_overlapped(launch, local: Sequence[torch.Tensor], n_frames: int, k: int, group=None, hooks=None):
    """Temporal NLM of this rank's block with the halo exchange hidden behind the interior frames.

    `launch(frames, first, count, out_offset)` filters outputs [first, first+count) of the ordered
    frame list `frames` (what mid_nlm_temporal takes) and stores them at block-relative position
    out_offset.  Order of events: post the halo transfers -> launch the interior outputs (their
    windows lie inside the block, no halo needed) -> wait for the transfers -> launch the <= 2k
    boundary outputs.  Returns the `have` dict (frames by global id).

    Streams (device backends): `launch` must enqueue on the current torch stream, see _order_after_halo.
    hooks (optional dict): 'stats' -> dict that receives halo_bytes_recv / halo_bytes_sent / halo_frames_recv;
    'before_wait' / 'after_wait' -> callables run around the wait for the transfers (bench.py records stream
    events there to time how long the boundary frames were held up by the halo)."""
    hooks = hooks or {}
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, count = partition(n_frames, world)[rank]
    have, reqs = start_halo_exchange(local, n_frames, k, group)
    if "stats" in hooks:
        recv, send = halo_plan(n_frames, world, k, rank) if (world > 1 and k > 0 and count > 0) else ([], [])
        fb = local[0].numel() * local[0].element_size() if count else 0
        hooks["stats"].update(halo_frames_recv=sum(len(i) for _, i in recv), halo_bytes_recv=fb * sum(len(i) for _, i in recv),
                              halo_bytes_sent=fb * sum(len(i) for _, i in send))
    if count == 0:
        return have
    plan = block_launch_plan(n_frames, world, k, rank)
    for phase, w_lo, w_hi, first, cnt, off in plan:
        if phase == "interior":
            launch([have[f] for f in range(w_lo, w_hi + 1)], first, cnt, off)
    if "before_wait" in hooks:
        hooks["before_wait"]()
    for r in reqs:
        r.wait()
    if have.finalize:
        have.finalize()
    if reqs:
        _order_after_halo(local)
    if "after_wait" in hooks:
        hooks["after_wait"]()
    for phase, w_lo, w_hi, first, cnt, off in plan:
        if phase == "boundary":
            launch([have[f] for f in range(w_lo, w_hi + 1)], first, cnt, off)
    return have


def block_launch_plan(n_frames: int, world: int, k: int, rank: int):
    """The launches one rank makes for its block, as pure data: a list of
    (phase, w_lo, w_hi, first, count, out_offset) where frames w_lo..w_hi (global ids) form the table handed to
    mid_nlm_temporal, outputs are table entries [first, first+count), stored at block-relative out_offset.
    phase "interior": every window t-k..t+k (clipped at the SEQUENCE ends only) lies inside the rank's own block,
    so the launch needs no halo frame and can run while the halo is in flight; the table is restricted to the
    block so that clipping at the block edge can never stand in for a missing halo frame.
    phase "boundary": the <= 2k outputs next to a neighbouring block; their tables include the halo frames."""
    start, count = partition(n_frames, world)[rank]
    if count == 0:
        return []
    lo_int = start if start == 0 else start + k
    hi_int = start + count if start + count == n_frames else start + count - k      # exclusive
    plan = []
    if hi_int > lo_int:
        w_lo, w_hi = max(start, lo_int - k), min(start + count - 1, hi_int - 1 + k)
        plan.append(("interior", w_lo, w_hi, lo_int - w_lo, hi_int - lo_int, lo_int - start))
    for a, b in ((start, min(lo_int, start + count)), (max(hi_int, lo_int), start + count)):
        if b > a:
            w_lo, w_hi = max(0, a - k), min(n_frames - 1, b - 1 + k)
            plan.append(("boundary", w_lo, w_hi, a - w_lo, b - a, a - start))
    return plan


def window_for_block(have, n_frames: int, k: int, start: int, count: int):
    """Ordered frame list covering the block and its clipped halo, plus `first` = index of the
    block's first frame inside it: the arguments mid_nlm_temporal takes."""
    lo, hi = max(0, start - k), min(n_frames - 1, start + count - 1 + k)
    return [have[f] for f in range(lo, hi + 1)], start - lo


def temporal_nlm_block(ctx, have, n_frames, k, start, count, outs, hparam, search, patch, fmt=0, stream=None):
    """Runs the temporal NLM of this rank's block on device tensors (torch CUDA tensors or anything
    with data_ptr()); `outs` are `count` preallocated RGBA32F tensors.
    stream=None means the CURRENT torch stream for CUDA tensors -- the one exchange_halo()'s waits ordered -- and
    raises if that is the default stream (see launch_stream_for); pass an explicit handle to override."""
    if count == 0:
        return
    frames, first = window_for_block(have, n_frames, k, start, count)
    if stream is None:
        stream = launch_stream_for(frames[0])
    h, w = frames[0].shape[0], frames[0].shape[1]
    ctx.nlm_temporal_dev([f.data_ptr() for f in frames], [o.data_ptr() for o in outs], w, h, hparam,
                         search, patch, k, first, count, fmt, stream)
