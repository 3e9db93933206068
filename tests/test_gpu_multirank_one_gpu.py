"""GPU suite: the N-rank paths with REAL kernels on a one-GPU box.  RCCL refuses two ranks on one device, so the ranks
share cuda:0, the process group is gloo and the halo frames are staged through host memory (sharding.start_halo_exchange
does that by itself for gloo + device tensors) -- everything else is the code the multi-GPU runs execute: partition,
launch plan, interior-before-halo ordering, the stream rule, bench.py's rank launcher, barriers, max-over-ranks,
all_gather of the per-rank halo statistics.  What it cannot cover is the RCCL/xGMI transport itself."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import image_denoising_filter_amd as mid
from image_denoising_filter_amd import sharding
rank, world, n, k, h, w = (int(x) for x in sys.argv[2:8])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = mid.Context(0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)          # the stream rule: a non-default stream is current
rng = np.random.default_rng(4242)
seq = [(rng.random((h, w, 4), dtype=np.float32) * 0.8).astype(np.float32) for _ in range(n)]   # same on every rank
start, count = sharding.partition(n, world)[rank]
local = [torch.from_numpy(seq[start + i]).to(dev) for i in range(count)]
outs = [torch.empty((h, w, 4), device=dev, dtype=torch.float32) for _ in range(count)]
done = []
def launch(frames, first, cnt, off):
    ctx.nlm_temporal_dev([f.data_ptr() for f in frames], [o.data_ptr() for o in outs[off:off + cnt]], w, h, 0.5, (-10, 11), (-3, 4),
                         k, first, cnt, mid.FMT_RGBA32F, sharding.launch_stream_for(frames[0]))
    done.extend(range(off, off + cnt))
stats = {}
have = sharding.temporal_block_overlapped(launch, local, n, k, hooks={"stats": stats})
torch.cuda.synchronize()
assert sorted(done) == list(range(count)), done
lo, hi = max(0, start - k), min(n - 1, start + count - 1 + k)
assert sorted(have) == list(range(lo, hi + 1))
assert all(t.is_cuda for t in have.values())
for f, t in have.items():
    assert np.array_equal(t.cpu().numpy(), seq[f]), f
dist.barrier()
pickle.dump((rank, start, [o.cpu().numpy() for o in outs], stats), open(sys.argv[8] + f".{rank}", "wb"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,n,k", [(2, 10, 2), (3, 11, 2), (4, 8, 1)])
def test_sharded_temporal_nlm_with_real_kernels_equals_one_shard(ctx, tmp_path, world, n, k):
    import pickle
    h, w = 70, 130
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    base = str(tmp_path / "out")
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(n), str(k), str(h), str(w), base],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, f"rank {r}:\n{out[-3000:]}"
    rng = np.random.default_rng(4242)
    seq = [(rng.random((h, w, 4), dtype=np.float32) * 0.8).astype(np.float32) for _ in range(n)]
    whole = ctx.nlm_temporal(seq, k=k, search=(-10, 11), patch=(-3, 4))
    seen = 0
    frame_bytes = h * w * 16
    for r in range(world):
        rank, start, outs, stats = pickle.load(open(base + f".{r}", "rb"))
        for i, o in enumerate(outs):
            assert np.array_equal(o, whole[start + i]), f"rank {rank} frame {start + i} differs from the one-shard result"
            seen += 1
        interior = 0 < r < world - 1
        assert stats["halo_bytes_recv"] == frame_bytes * k * (2 if interior else 1)
    assert seen == n


def test_bench_two_ranks_rehearsal_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher: it starts its own two ranks; MID_BENCH_REHEARSE=1 lets them share the
    one GPU of this box (gloo, host-staged halo).  The line must be rank 0's, say n_gpus 2, be flagged as a rehearsal,
    and carry the per-rank halo statistics of the 64-frame temporal sequence (32 frames per rank, 2 halo frames each)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MID_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["value"] > 0 and d["scaling"] == "weak"
    t = d["also"]["temporal_nlm_k2"]
    assert t["frames"] == 64 and t["frames_per_rank"] == [32, 32]
    assert t["halo_bytes_recv_per_rank"] == [2 * 1920 * 1080 * 16] * 2 and t["halo_bytes_sent_per_rank"] == [2 * 1920 * 1080 * 16] * 2
    assert len(t["halo_held_ms_per_rank"]) == 2 and t["Mpixel/s_out"] > 0
    assert not [k for k in d["also"] if k.endswith("_error")], d["also"]
    pg = d["config"]["process_group"]                   # what the ranks ran on, in the line itself
    assert pg["backend"] == "gloo" and len(pg["devices"]) == 2 and all("gfx950" in n or "MI355" in n or "cuda:0" in n for n in pg["devices"])
    assert pg["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
