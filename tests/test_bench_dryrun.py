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
