"""CPU suite, part 6: bench.py's multi-rank control flow rehearsed under torch.distributed.run with gloo
(`--dry-run`: no kernels, tiny frames) -- process group, barriers, max-over-ranks, the overlapped halo
exchange of the temporal step, one JSON line from rank 0.  The real N>1 runs are the driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [1, 2, 4])
def test_bench_control_flow(n):
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--dry-run"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
               "--frames", "4", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == n and d["value"] is None


@pytest.mark.parametrize("n", [2, 4])
def test_bench_spawns_its_own_ranks(n):
    """`python bench.py --gpus N` with NO launcher and no RANK/WORLD_SIZE in the environment: the parent starts N
    ranks itself (before anything touches a GPU), relays rank 0's one JSON line, and n_gpus is the process
    group's world size."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--frames", "4", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == n


def test_bench_launcher_reports_a_failed_rank(tmp_path):
    """A rank that dies makes the launcher stop the others and return non-zero -- never a 1-GPU number.  bench.py has no
    failure hook of its own: the launcher is driven through a wrapper script whose rank 1 exits (code 7) the moment the
    rehearsal builds its stand-in context, i.e. after the process group is up and before the first barrier."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    wrapper = tmp_path / "bench.py"
    wrapper.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import importlib.util\n"
        f"spec = importlib.util.spec_from_file_location('bench_under_test', {os.path.join(ROOT, 'bench.py')!r})\n"
        "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
        "bench.__file__ = os.path.abspath(__file__)          # launch_ranks() re-starts THIS script for every rank\n"
        "def die(self):\n"
        "    if os.environ.get('RANK') == '1':\n"
        "        os._exit(7)\n"
        "    self.launches = []\n"
        "bench._DryContext.__init__ = die\n"
        "bench.main()\n")
    cmd = [sys.executable, str(wrapper), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "4", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_mismatched_launcher():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
