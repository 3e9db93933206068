"""GPU suite: the library's ROCTx ranges (csrc/markers.cpp) as a profiler sees them.

The reference brackets every submit with timestamp queries (src/main.cpp:793-796,812-814,842-844); the library's counterpart for a
TRACE are named host ranges around the same stages.  They bind to a ROCTx the process already holds -- `rocprofv3 --marker-trace`
preloads librocprofiler-sdk-roctx -- and are inert otherwise.  Checked here: without a profiler mid_range_push says so (0) and
loads nothing; under `rocprofv3 --marker-trace` one pipeline call yields one call range with exactly one `upload f`, `nlm t`,
`download t` range per frame nested inside it, in issue order, and the caller's own ranges (mid_range_push / mid_range_pop)
around it."""
import csv
import glob
import os
import shutil
import subprocess
import sys

import pytest

import image_denoising_filter_amd as mid
from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
frames = [rng.random((96, 160, 4), dtype=np.float32) for _ in range(5)]
assert mid.lib.mid_range_push(b"caller stage") == 1          # a ROCTx is present under the profiler
outs, _ = ctx.sequence_nlm(frames, k=1)
assert mid.lib.mid_range_pop() == 1
whole = ctx.nlm_temporal(frames, k=1)
assert all(np.array_equal(a, b) for a, b in zip(outs, whole))
print("MARKERS done")
'''


def test_ranges_are_inert_without_a_profiler():
    def roctx_libs():
        return sorted({l.split()[-1] for l in open("/proc/self/maps") if "roctx" in l})
    before = roctx_libs()                                        # (torch maps its own libroctx64 privately: not in the global scope)
    assert mid.lib.mid_range_push(b"nobody listens") == 0
    assert mid.lib.mid_range_pop() == 0
    assert roctx_libs() == before                                # nothing was loaded for it


def test_pipeline_ranges_under_rocprofv3_marker_trace(tmp_path):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        pytest.fail("rocprofv3 is part of the image; not found")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run([rocprof, "--marker-trace", "--output-format", "csv", "-d", str(tmp_path / "prof"), "--",
                        sys.executable, str(script), ROOT], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MARKERS done" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    traces = glob.glob(str(tmp_path / "prof" / "**" / "*marker_api_trace.csv"), recursive=True)
    assert len(traces) == 1, traces
    rows = list(csv.DictReader(open(traces[0])))
    name_col = "Function" if "Function" in rows[0] else "Message"
    rng_ = [(r_[name_col], int(r_["Start_Timestamp"]), int(r_["End_Timestamp"])) for r_ in rows]
    call = [x for x in rng_ if x[0].startswith("mid_sequence_nlm frames=5 k=1 outputs=[0,5)")]
    assert len(call) == 1, [x[0] for x in rng_]
    _, c0, c1 = call[0]
    inside = sorted((x for x in rng_ if c0 <= x[1] and x[2] <= c1 and x is not call[0]), key=lambda x: x[1])
    names = [x[0] for x in inside]
    for stage in ("upload", "nlm", "download"):
        assert sorted(n for n in names if n.startswith(stage + " ")) == [f"{stage} {i}" for i in range(5)], names
    assert names.count("drain") == 1 and names[-1] == "drain"
    # issue order: a frame's launch comes after the uploads its window needs, its download after its launch
    pos = {n: i for i, n in enumerate(names)}
    for t in range(5):
        assert pos[f"upload {min(t + 1, 4)}"] < pos[f"nlm {t}"] < pos[f"download {t}"], names
    outer = [x for x in rng_ if x[0] == "caller stage"]
    assert len(outer) == 1 and outer[0][1] <= c0 and c1 <= outer[0][2]
