"""GPU suite: the C++ RCCL path (csrc/sharded.cpp) on the one-GPU box.  A 1-rank communicator through the C-ABI
(mid_comm_unique_id -> mid_comm_create: ncclGetUniqueId / ncclCommInitRank bound by dlopen), in a fresh child process:
  * mid_comm_loopback: ncclRecv + ncclSend addressed to this rank inside one group on the exchange stream, a whole
    1080p RGBA32F frame -- the halo exchange's call pattern, minus the wire;
  * mid_nlm_temporal_sharded with world = 1 == mid_nlm_temporal over the whole sequence, bit for bit (launch plan,
    stream/event plumbing, receive-buffer bookkeeping all run; nothing is exchanged);
  * mi_denoise --animation --halo rccl (ncclCommInitAll, one device) writes the files the default --halo host writes.
NOT covered here (needs >= 2 devices): two ranks, the xGMI transport, overlap of the halo with the interior launches.
The plans for every world size are pinned on the CPU (tests/test_shard_native_plan.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import image_denoising_filter_amd as mid
from conftest import ROOT, synth_hdr

pytestmark = pytest.mark.gpu

_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
uid = mid.comm_unique_id()
assert len(uid) == 128 and any(uid)
rep = {}
with mid.Comm(ctx, uid, 0, 1) as comm:
    # a frame-sized send/receive to self
    h, w = 1080, 1920
    rng = np.random.default_rng(5)
    frame = rng.random((h, w, 4), dtype=np.float32)
    src, dst = ctx.upload(frame), ctx.zeros(frame.nbytes)
    try:
        comm.last_loopback()
        raise SystemExit("a loopback timeline before any loopback")
    except mid.MidError as e:
        assert e.code == 1, str(e)
    comm.loopback(src.ptr, dst.ptr, frame.nbytes)
    ctx.sync()
    assert np.array_equal(ctx.download(dst, frame.shape, np.float32), frame)
    rep["loopback_bytes"] = frame.nbytes
    t0, t1 = comm.last_loopback()
    assert 0.0 <= t0 < t1 < 1000.0, (t0, t1)
    rep["loopback_ms"] = [t0, t1]
    # the sharded entry point with one rank
    h, w, n, k = 70, 130, 7, 2
    seq = [(rng.random((h, w, 4), dtype=np.float32) * 0.8).astype(np.float32) for _ in range(n)]
    d_in = [ctx.upload(f) for f in seq]
    d_out = [ctx.alloc(h * w * 16) for _ in seq]
    for rep_i in range(2):          # twice: the second call reuses the communicator's buffers and events
        comm.nlm_temporal_sharded_dev([d.ptr for d in d_in], [d.ptr for d in d_out], w, h, n, k, 0.5, (-10, 11), (-3, 4), mid.FMT_RGBA32F)
    ctx.sync()
    whole = ctx.nlm_temporal(seq, k=k, search=(-10, 11), patch=(-3, 4))
    for i in range(n):
        assert np.array_equal(ctx.download(d_out[i], (h, w, 4), np.float32), whole[i]), i
    rep["last_exchange"] = comm.last_exchange()
    # a loopback afterwards has its own events: the sharded call's timeline reads the same before and after it
    tl = comm.last_timeline()
    comm.loopback(src.ptr, dst.ptr, frame.nbytes)
    ctx.sync()
    assert comm.last_timeline() == tl and tl["end_ms"] > 0.0 and 0.0 < tl["interior_end_ms"] <= tl["end_ms"], tl
    rep["timeline_kept"] = True
    # argument errors surface as codes, not crashes
    try:
        comm.nlm_temporal_sharded_dev([d.ptr for d in d_in[:3]], [d.ptr for d in d_out[:3]], w, h, n, k, 0.5, (-10, 11), (-3, 4), 0)
        raise SystemExit("short block accepted")
    except ValueError:
        pass
    # the stream rule: the receive buffers are reused from call to call, ordered only through the stream the calls are
    # issued on -- a call on ANOTHER stream while the previous one is in flight is refused, after a synchronisation accepted
    import torch
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    hb, wb, nb = 1080, 1920, 8
    big_in = [ctx.upload(frame) for _ in range(nb)]
    big_out = [ctx.alloc(frame.nbytes) for _ in range(nb)]
    args = ([d.ptr for d in big_in], [d.ptr for d in big_out], wb, hb, nb, 2, 0.5, (-10, 11), (-3, 4), mid.FMT_RGBA32F)
    comm.reserve(frame.nbytes, 2)
    with torch.cuda.stream(sA):
        torch.cuda._sleep(100_000_000)                                   # stream A is busy for 40 ms (shader clock) to 1 s (100 MHz counter),
    comm.nlm_temporal_sharded_dev(*args, stream=sA.cuda_stream)          # so this call is deterministically still in flight ...
    try:
        comm.nlm_temporal_sharded_dev(*args, stream=sB.cuda_stream)      # ... when this one arrives on the other stream
        raise SystemExit("a second stream was accepted while the first call was in flight")
    except mid.MidError as e:
        assert e.code == 1 and "another stream" in str(e), str(e)
    comm.nlm_temporal_sharded_dev(*args, stream=sA.cuda_stream)          # the same stream again: fine
    ctx.sync(sA.cuda_stream)
    comm.nlm_temporal_sharded_dev(*args, stream=sB.cuda_stream)          # after the synchronisation: fine
    ctx.sync(sB.cuda_stream)
    one = ctx.nlm_temporal([frame] * 3, k=2, first=0, count=1, search=(-10, 11), patch=(-3, 4))[0]
    assert np.array_equal(ctx.download(big_out[0], frame.shape, np.float32), one)
    rep["stream_rule"] = "refused, then accepted"
    # abort: the handle then only accepts destroy
    comm.abort()
    try:
        comm.nlm_temporal_sharded_dev(*args)
        raise SystemExit("an aborted communicator accepted a call")
    except mid.MidError as e:
        assert e.code == 1 and "aborted" in str(e), str(e)
    rep["abort"] = "refused afterwards"
print("SHARD1 " + json.dumps(rep), flush=True)
'''


def test_one_rank_communicator_loopback_and_sharded_temporal_nlm(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("SHARD1 ")][0][7:])
    assert rep["loopback_bytes"] == 1920 * 1080 * 16
    assert rep["last_exchange"] == [0, 0, 0.0]                     # one rank: nothing to exchange
    assert rep["timeline_kept"] and rep["loopback_ms"][1] > rep["loopback_ms"][0]
    assert rep["stream_rule"] == "refused, then accepted" and rep["abort"] == "refused afterwards"


@pytest.mark.parametrize("hdr", [True, False])
def test_cli_animation_with_rccl_halo_equals_host_halo(tmp_path, hdr):
    from test_cli import CLI, _make_animation
    d, frames, _, ext = _make_animation(tmp_path, hdr, n=6)
    outs = {}
    for mode in ("host", "rccl"):
        o = tmp_path / mode
        o.mkdir()
        r = subprocess.run([CLI, str(d / f"Animation01_X_0000.{ext}"), "--animation", "--temporal-k", "2", "--outdir", str(o), "--halo", mode],
                           cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[mode] = [mid.load_image(o / f"output-animation-Animation01_X_{i:04d}.{ext}") for i in range(6)]
    for a, b in zip(outs["host"], outs["rccl"]):
        if hdr:
            assert np.array_equal(a, b)
        else:   # host: u8 conversion in the NLM kernel's epilogue; rccl: mid_pack_u8 of the float result -- the same function of the same floats
            assert np.array_equal(a, b)
