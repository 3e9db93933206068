"""GPU suite: the first contact with RCCL on hardware, on the one-GPU box.

A 1-rank `nccl` process group (backend "nccl" IS RCCL on ROCm) in a fresh child process -- environment set before
anything touches the GPU -- doing exactly what bench.py does with a group: init_process_group(device_id=...), barrier,
all_reduce(MAX) on a device tensor, all_gather, all_gather_object, then sharding.temporal_block_overlapped under it with
the stream rule, compared bit for bit with the whole-sequence launch.

Covered: librccl loads, the torch build's nccl backend initialises a communicator on the MI355X, its collectives run on
device tensors, the sharded control flow runs under a device backend (not gloo) on a non-default stream, and a
self-addressed isend/irecv pair (ncclSend/ncclRecv inside one group) moves a frame-sized device buffer.
NOT covered (needs >= 2 devices, which this pool never gives a builder): the xGMI peer-to-peer transport between two
GPUs and any scaling figure.
"""
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
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
mode = sys.argv[2]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
report = {"backend": str(dist.get_backend()), "world": dist.get_world_size(), "device": torch.cuda.get_device_name(0)}
dist.barrier()
t = torch.tensor([3.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
g = [torch.empty(4, device=dev, dtype=torch.float64)]
dist.all_gather(g, torch.arange(4, device=dev, dtype=torch.float64))
assert g[0].tolist() == [0.0, 1.0, 2.0, 3.0]
names = [None]
dist.all_gather_object(names, "cuda:0 test")
assert names == ["cuda:0 test"]

import image_denoising_filter_amd as mid
from image_denoising_filter_amd import sharding
ctx = mid.Context(0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)          # the stream rule: a non-default stream is current
h, w, n, k = 70, 130, 7, 2
rng = np.random.default_rng(99)
seq = [(rng.random((h, w, 4), dtype=np.float32) * 0.8).astype(np.float32) for _ in range(n)]
local = [torch.from_numpy(f).to(dev) for f in seq]
outs = [torch.empty((h, w, 4), device=dev, dtype=torch.float32) for _ in range(n)]
def launch(frames, first, cnt, off):
    ctx.nlm_temporal_dev([f.data_ptr() for f in frames], [o.data_ptr() for o in outs[off:off + cnt]], w, h, 0.5, (-10, 11), (-3, 4),
                         k, first, cnt, mid.FMT_RGBA32F, sharding.launch_stream_for(frames[0]))
stats = {}
have = sharding.temporal_block_overlapped(launch, local, n, k, hooks={"stats": stats})
torch.cuda.synchronize()
assert sorted(have) == list(range(n)) and stats["halo_bytes_recv"] == 0
whole = ctx.nlm_temporal(seq, k=k, search=(-10, 11), patch=(-3, 4))
for i in range(n):
    assert np.array_equal(outs[i].cpu().numpy(), whole[i]), i
report["sharded_under_nccl"] = "bit-identical to the one-launch result"

if mode == "p2p":
    # ncclSend + ncclRecv addressed to this very rank inside one group: the halo exchange's call pattern, minus the wire
    src = torch.from_numpy(seq[0]).to(dev)
    dst = torch.zeros_like(src)
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, 0), dist.P2POp(dist.isend, src, 0)])
    for r in reqs:
        r.wait()
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    report["self_p2p"] = "ok"
dist.barrier()
dist.destroy_process_group()
print("RCCL1 " + json.dumps(report), flush=True)
'''


def _run(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # as bench.py sets it (dmabuf IPC on this pool's driver)
    r = subprocess.run([sys.executable, str(script), ROOT, mode], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL1 ")]
    assert len(line) == 1, r.stdout[-2000:]
    return json.loads(line[0][6:])


def test_one_rank_nccl_group_collectives_and_sharded_temporal_nlm(tmp_path):
    rep = _run(tmp_path, "collectives")
    assert rep["backend"] == "nccl" and rep["world"] == 1
    assert rep["sharded_under_nccl"].startswith("bit-identical")


def test_one_rank_nccl_self_send_recv_of_a_frame(tmp_path):
    rep = _run(tmp_path, "p2p")
    assert rep["self_p2p"] == "ok"


def test_bench_line_under_a_launcher_with_nccl_names_backend_and_devices(tmp_path):
    """bench.py as the driver starts it for N > 1 -- RANK/WORLD_SIZE from a launcher, backend nccl -- cannot run with two
    ranks here; what can: the 1-GPU line (no group), which must not claim one."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "2",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and "process_group" not in d["config"]
    assert d["roofline"]["kernel_ms_min"] <= d["roofline"]["avg_launch_ms"] <= d["roofline"]["kernel_ms_max"]


def test_bench_stdout_is_exactly_one_json_line_with_the_native_rccl_extra(tmp_path):
    """The whole 1-GPU bench with its extras -- among them the C++ RCCL path (a 1-rank communicator created through the
    C-ABI, which makes RCCL print its version banner on stdout): stdout must still hold the JSON line and nothing else,
    and the native path must report outputs bit-identical to the plain launch."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    d = json.loads(lines[0])
    nat = d["also"]["temporal_nlm_k2_native"]
    assert nat["bit_identical_to_single_launch_per_rank"] == [True] and nat["halo_bytes_recv_per_rank"] == [0]
    assert not [k for k in d["also"] if k.endswith("_error")], d["also"]
    assert d["also"]["bilateral_r8_texture_over_linear"] > 0
    # the sharded path at the top level of the line: configs[4] through the C++ RCCL path, with what the
    # communicator itself reports; one rank here, so nothing was exchanged and the line says so
    ss = d["scaling_strong"]
    assert ss["n_gpus"] == 1 and ss["scaling"] == "strong" and ss["value"] == nat["Mpixel/s_out"] > 0
    assert ss["bit_identical_to_single_launch_per_rank"] == [True] and ss["bytes_on_the_wire"] == 0
    assert ss["rccl_comm_count"] == [1] and ss["rccl_version"] > 0 and ss["halo_hidden_frac"] is None
    assert "unmeasured on hardware" in ss["hardware_status"]
    assert nat["issue_order_rank0"].startswith("I") and "X" not in nat["issue_order_rank0"]
    pr = nat["exchange_stream_priority"]
    assert pr["priority"] == pr["greatest"] <= pr["least"]
    assert nat["timeline_ms_per_rank"][0]["end"] > 0 and nat["timeline_ms_per_rank"][0]["exchange_end"] == 0
    # the contract's two-valued enum, with the truthful value beside it
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["bound_actual"] == "valu"
    assert len(d["roofline"]["kernel_code_sha256"]) == 64


def test_bench_watchdog_prints_the_headline_names_the_hung_extra_and_exits_nonzero(tmp_path):
    """An extra that never returns (here simulated; on N > 1 it would be a collective whose peer is gone): the watchdog
    prints the complete headline line with `also.error` naming the extra in flight and ends the process with exit code 3 --
    for every world size (a process abandoned in the middle of GPU work must not report success)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # bench.py carries no test hook: this wrapper imports it, shortens its watchdog and makes the one library call that only the
    # `bilateral_batch` extra uses (mid_bilateral_batch through Context.bilateral_batch_dev) never return
    wrapper = tmp_path / "bench_with_a_hung_extra.py"
    wrapper.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "import image_denoising_filter_amd as mid\n"
        "bench.WATCHDOG_S = 25\n"
        "mid.Context.bilateral_batch_dev = lambda *a, **k: time.sleep(3600)\n"
        "sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--frames', '2', '--no-cpu-baseline']\n"
        "bench.main()\n")
    r = subprocess.run([sys.executable, str(wrapper)], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and "bilateral_batch" in d["also"]["error"] and "exit code 3" in d["also"]["error"]
    assert "bilateral_r8_linear" in d["also"], "extras that finished before the hang are in the line"
