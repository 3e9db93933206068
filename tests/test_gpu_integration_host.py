"""GPU suite: INTEGRATION.md's binding, executed.  tests/integration_host.cpp is a C++20 host program -- the reference's language
-- with the call sequence INTEGRATION.md gives for the body of RunOnGPU (src/main.cpp:1307-1730), fed from std::vectors like the
reference's imageData / layerData / resultHDRData (pageable memory: the library's bounce buffers carry it).  Its outputs must be
the bytes the same entry points produce when Python drives them: the document's sequence is the API's real contract, for the
plain bilateral in both addressings, the layer loop, the per-frame NLM accumulate + normalize, HDR and LDR (u8 pack)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, synth_hdr, synth_ldr

pytestmark = pytest.mark.gpu
H, W = 135, 240


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("integration")
    out = d / "integration_host"
    libdir = os.path.join(ROOT, "image_denoising_filter_amd")
    subprocess.run(["g++", "-std=c++20", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "integration_host.cpp"), "-o", str(out), "-L", libdir, "-lmi_denoise",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=300)
    return str(out)


def _run(exe, tmp_path, mode, hdr, img, extras=()):
    (tmp_path / "in.raw").write_bytes(np.ascontiguousarray(img).tobytes())
    names = []
    for i, e in enumerate(extras):
        (tmp_path / f"x{i}.raw").write_bytes(np.ascontiguousarray(e).tobytes())
        names.append(str(tmp_path / f"x{i}.raw"))
    r = subprocess.run([exe, mode, str(W), str(H), "1" if hdr else "0", str(tmp_path / "in.raw"), str(tmp_path / "out.raw")] + names,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = (tmp_path / "out.raw").read_bytes()
    return np.frombuffer(raw, np.float32 if hdr else np.uint8).reshape(H, W, 4)


@pytest.mark.parametrize("hdr", [True, False])
def test_the_documented_call_sequence_gives_the_bytes_of_the_python_driven_calls(ctx, exe, tmp_path, hdr):
    rng = np.random.default_rng(21 + hdr)
    frames = [synth_hdr(rng, H, W) * 0.3 if hdr else synth_ldr(rng, H, W) for _ in range(3)]
    layers = [synth_ldr(rng, H, W) for _ in range(2)]
    post = (lambda x: x) if hdr else ctx.pack_u8                          # GetImageFromGPU's u8 conversion for LDR (src/main.cpp:97-103)
    t = frames[0]
    assert np.array_equal(_run(exe, tmp_path, "bilateral", hdr, t), post(ctx.bilateral(t, 20, 2.0, 0.2, "texture")))
    assert np.array_equal(_run(exe, tmp_path, "linear", hdr, t), post(ctx.bilateral(t, 20, 2.0, 0.2, "linear")))
    assert np.array_equal(_run(exe, tmp_path, "layers", hdr, t, layers), post(ctx.bilateral_layers(t, layers, 20, 2.0, 0.2)))
    Wacc = np.zeros((H, W, 8), np.float32)
    for f in frames:                                                      # target first, then the other frames: one accumulate each
        Wacc = ctx.nlm_accum(t, f, Wacc, 0.5, (-7, 7), (-3, 3))
    assert np.array_equal(_run(exe, tmp_path, "nlm", hdr, t, frames[1:]), post(ctx.normalize(Wacc)))


def test_the_multi_gpu_snippet_with_one_rank_over_real_rccl(ctx, exe, tmp_path):
    """INTEGRATION.md "An animation over several GPUs": mid_comm_unique_id -> mid_comm_create -> mid_shard_block ->
    mid_nlm_temporal_sharded -> mid_comm_last_timeline, from C++, with a 1-rank RCCL communicator (all this pool can offer): the
    outputs are mid_nlm_temporal's over the whole sequence, and the timeline says "no exchange" (0, 0) and a positive end."""
    rng = np.random.default_rng(5)
    frames = [synth_hdr(rng, H, W) * 0.3 for _ in range(6)]
    (tmp_path / "in.raw").write_bytes(frames[0].tobytes())
    names = []
    for i, f in enumerate(frames[1:]):
        (tmp_path / f"f{i}.raw").write_bytes(f.tobytes())
        names.append(str(tmp_path / f"f{i}.raw"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe, "sharded", str(W), str(H), "1", str(tmp_path / "in.raw"), str(tmp_path / "out.raw")] + names,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    raw = np.frombuffer((tmp_path / "out.raw").read_bytes(), np.float32)
    outs, t = raw[:-4].reshape(6, H, W, 4), raw[-4:]
    want = ctx.nlm_temporal(frames, k=2, search=(-7, 7), patch=(-3, 3))
    assert all(np.array_equal(outs[i], want[i]) for i in range(6))
    assert t[0] == 0 and t[1] == 0 and t[3] > 0 and t[2] <= t[3]


def test_a_missing_input_is_a_runtime_error_and_exit_failure(exe, tmp_path):
    r = subprocess.run([exe, "bilateral", "8", "8", "1", str(tmp_path / "nope.raw"), str(tmp_path / "o.raw")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "unexpected size" in r.stderr
