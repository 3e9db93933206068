"""The drop-in CLI `mi_denoise`: same positional argument, mode order, output
names and timing lines as the reference's main() (src/main.cpp:1935-1994).

CPU part: the two "Running on CPU" modes (the reference's own RunOnCPU feature) against the oracle
restatement that is pinned to the reference loop -- bit-exact, PNG and EXR.
GPU part: all six GPU modes on a small synthetic animation directory (frames + RenderElements
layers) against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import image_denoising_filter_amd as mid
import oracle
from conftest import ROOT, rel_err, synth_hdr, synth_ldr

CLI = os.path.join(ROOT, "image_denoising_filter_amd", "mi_denoise")


def _run(args, cwd):
    return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True, timeout=600)


def _make_animation(root, hdr, n=4, h=40, w=56):
    """<root>/Anim/Animation01_XXX_000N.{png,exr} + RenderElements/{albedo,normal}_000N.png"""
    rng = np.random.default_rng(7)
    d = root / "Anim"
    (d / "RenderElements").mkdir(parents=True)
    ext = "exr" if hdr else "png"
    frames = []
    base = synth_hdr(rng, h, w, 1.0) * 0.25
    for i in range(n):
        if hdr:
            f = (np.roll(base, 2 * i, axis=1) * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32)
        else:
            f = synth_ldr(rng, h, w)
        frames.append(f)
        mid.save_image(d / f"Animation01_X_{i:04d}.{ext}", f)
    layers = {}
    for i in range(n):
        layers[i] = [synth_ldr(rng, h, w) for _ in range(2)]
        mid.save_image(d / "RenderElements" / f"albedo_{i:04d}.png", layers[i][0])
        mid.save_image(d / "RenderElements" / f"normal_{i:04d}.png", layers[i][1])
    (d / "notes.txt").write_text("not an image")
    return d, frames, layers, ext


@pytest.mark.parametrize("hdr", [False, True])
def test_cpu_modes_match_the_pinned_oracle(tmp_path, hdr):
    d, frames, _, ext = _make_animation(tmp_path, hdr, n=1, h=33, w=41)
    out = tmp_path / "out"
    out.mkdir()
    r = _run([str(d / f"Animation01_X_0000.{ext}"), "--cpu-only", "--outdir", str(out), "--cpu-threads", "1,3"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("Running on CPU") == 2 and "Running on CPU (1 thread bialteral)" in r.stdout
    assert "Running on CPU (3 threads bialteral)" in r.stdout and r.stdout.count("Time taken: ") == 2
    got = mid.load_image(out / f"output-cpu.{ext}")
    src = frames[0] if hdr else oracle.unpack_u8(frames[0], flavour=1)           # CPU decode c*(1/255), src/main.cpp:1804
    ref = oracle.cpu_bilateral(src, 10, 10.0, 0.2, blue_bug=True, threads=1)
    if hdr:
        assert np.array_equal(got, ref)
    else:
        assert np.array_equal(got, oracle.pack_u8(ref))                           # truncating pack, src/main.cpp:1905-1911


def test_cpu_mode_options_and_errors(tmp_path):
    d, frames, _, ext = _make_animation(tmp_path, True, n=1, h=21, w=25)
    r = _run([str(d / "Animation01_X_0000.exr"), "--cpu-only", "--cpu-radius", "4", "--cpu-sigma-s", "3", "--cpu-fix-blue",
              "--cpu-threads", "2"], tmp_path)
    assert r.returncode == 0
    got = mid.load_image(tmp_path / "output-cpu.exr")                              # default outdir = CWD, like the reference
    assert np.array_equal(got, oracle.cpu_bilateral(frames[0], 4, 3.0, 0.2, blue_bug=False, threads=1))
    r = _run([str(tmp_path / "nope.png"), "--cpu-only"], tmp_path)
    assert r.returncode == 1 and "cannot open" in r.stdout                        # runtime_error -> message + EXIT_FAILURE
    r = _run(["--bogus"], tmp_path)
    assert r.returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("hdr", [False, True])
def test_all_gpu_modes_against_oracle(tmp_path, hdr):
    d, frames, layers, ext = _make_animation(tmp_path, hdr)
    out = tmp_path / "out"
    out.mkdir()
    target = 1
    r = _run([str(d / f"Animation01_X_{target:04d}.{ext}"), "--gpu-only", "--outdir", str(out), "--radius", "8"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    banners = [l for l in r.stdout.splitlines() if l.startswith("Running on GPU")]
    assert banners == ["Running on GPU (nonlinear bialteral)", "Running on GPU (nonlinear bialteral + layers)",
                       "Running on GPU (linear bialteral)", "Running on GPU (nonlocal)",
                       "Running on GPU (multiframe nonlocal)", "Running on GPU (multiframe nonlocal + overlapping)"]
    assert r.stdout.count("transfer time: ") == 6 and r.stdout.count("execution time: ") == 6
    names = ["output-nonlinear-bialteral", "output-nonlinear-bialteral-layers", "output-linear-bialteral",
             "output-nonlinear-nlm", "output-nonlinear-nlm-multiframe", "output-nonlinear-nlm-multiframe-overlap"]
    assert sorted(os.listdir(out)) == sorted(f"{n}.{ext}" for n in names)

    f32 = [f if hdr else oracle.unpack_u8(f, 0) for f in frames]                  # GPU path decodes UNORM c/255
    t = f32[target]
    h, w = t.shape[:2]
    Z = np.zeros((h, w, 8), np.float32)
    Wl = Z
    for l in layers[target]:
        Wl = oracle.bilateral_layers_accum(t, l, Wl, 8, 2.0, 0.2)
    # the reference's frame list: target first, then every sibling frame (target again among them)
    lst = [t] + f32
    Wm = Z
    for f in lst:
        Wm = oracle.nlm_accum(t, f, Wm, 0.5)
    expect = {
        names[0]: oracle.bilateral_texture(t, 8, 2.0, 0.2),
        names[1]: oracle.normalize(Wl),
        names[2]: oracle.bilateral_linear(t, 8, 2.0, 0.2),
        names[3]: oracle.normalize(oracle.nlm_accum(t, t, Z, 0.5)),
        names[4]: oracle.normalize(Wm),
        names[5]: oracle.normalize(Wm),                                           # <= 9 frames: same list as non-overlap
    }
    for n, ref in expect.items():
        got = mid.load_image(out / f"{n}.{ext}")
        if hdr:
            assert rel_err(got, ref) < 2e-5, n
        else:                                                                      # truncating u8 pack: <= 1 LSB at boundaries
            exp8 = oracle.pack_u8(ref).astype(np.int16)
            diff = np.abs(got.astype(np.int16) - exp8)
            # alpha = sum(w*1)/sum(w) sits exactly on the 254/255 truncation boundary: fp32 rounding order
            # decides (in the reference too), so it only has to be within 1 LSB; colour >= 99.9 % identical
            assert diff.max() <= 1 and (diff[..., :3] != 0).mean() < 1e-3, n


@pytest.mark.gpu
def test_gpu_temporal_window_option(tmp_path):
    d, frames, _, ext = _make_animation(tmp_path, True, n=5)
    r = _run([str(d / "Animation01_X_0002.exr"), "--gpu-only", "--modes", "multiframe", "--temporal-k", "1",
              "--search", "-10,11", "--patch", "-3,4"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    got = mid.load_image(tmp_path / "output-nonlinear-nlm-multiframe.exr")
    ref = oracle.nlm_temporal(frames, k=1, first=2, count=1, search=(-10, 11), patch=(-3, 4))[0]
    assert rel_err(got, ref) < 2e-5


@pytest.mark.gpu
def test_animation_mode(tmp_path):
    """--animation: every frame filtered with temporal NLM, frame blocks over devices (1 here; --gpus 3 on a
    1-GPU box must fail loudly, not fall back)."""
    d, frames, _, ext = _make_animation(tmp_path, True, n=6)
    out = tmp_path / "o"
    out.mkdir()
    r = _run([str(d / "Animation01_X_0000.exr"), "--animation", "--temporal-k", "2", "--outdir", str(out)], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = oracle.nlm_temporal(frames, k=2)
    for i in range(6):
        got = mid.load_image(out / f"output-animation-Animation01_X_{i:04d}.exr")
        assert rel_err(got, ref[i]) < 2e-5, i
    # --pinned-mb: page-locked memory is budgeted.  0 = every frame pageable; a budget of two frame pairs (frames are
    # 40x56x16 B = 35,840 B, so 1 MiB holds them all -- use the 0 case and the default to bracket) must give the same files
    out2 = tmp_path / "o2"
    out2.mkdir()
    r = _run([str(d / "Animation01_X_0000.exr"), "--animation", "--temporal-k", "2", "--outdir", str(out2), "--pinned-mb", "0"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "6 frame(s) beyond the page-locked budget" in r.stdout
    for i in range(6):
        name = f"output-animation-Animation01_X_{i:04d}.exr"
        assert np.array_equal(mid.load_image(out2 / name), mid.load_image(out / name)), i
    import torch
    if torch.cuda.device_count() == 1:
        r = _run([str(d / "Animation01_X_0000.exr"), "--animation", "--gpus", "3", "--outdir", str(out)], tmp_path)
        assert r.returncode == 1 and "device" in r.stdout


@pytest.mark.gpu
def test_animation_mode_ldr_frames_come_back_as_u8(tmp_path):
    """PNG animation: the u8 read-back conversion runs on the device (mid_sequence_nlm_range_u8); the PNGs must hold
    (unsigned char)(255*v) of the temporal filter, up to the one-LSB truncation boundary of a 1e-5 float difference."""
    d, frames, _, ext = _make_animation(tmp_path, False, n=5)
    out = tmp_path / "o"
    out.mkdir()
    r = _run([str(d / "Animation01_X_0000.png"), "--animation", "--temporal-k", "1", "--outdir", str(out)], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "decoded 5 frames" in r.stdout and "file(s) at a time" in r.stdout and "encoded 5 frames" in r.stdout
    assert r.stdout.count("encoding png") == 5
    # one file at a time (--io-threads 1) writes the same bytes as one file per worker thread
    out1 = tmp_path / "o1"
    out1.mkdir()
    r1 = _run([str(d / "Animation01_X_0000.png"), "--animation", "--temporal-k", "1", "--outdir", str(out1), "--io-threads", "1"], tmp_path)
    assert r1.returncode == 0 and "(1 file(s) at a time)" in r1.stdout, r1.stdout + r1.stderr
    for i in range(5):
        name = f"output-animation-Animation01_X_{i:04d}.png"
        assert (out1 / name).read_bytes() == (out / name).read_bytes(), i
    ref = oracle.nlm_temporal([oracle.unpack_u8(f, 0) for f in frames], k=1)
    for i in range(5):
        got = mid.load_image(out / f"output-animation-Animation01_X_{i:04d}.png")
        want = oracle.pack_u8(ref[i])
        assert got.dtype == np.uint8 and got.shape == want.shape
        diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and np.mean(diff[..., :3] == 0) >= 0.999, i


@pytest.mark.gpu
def test_sequence_range_blocks_equal_whole(ctx):
    """mid_sequence_nlm_range over blocks == the whole sequence (the unit of frame-block sharding)."""
    rng = np.random.default_rng(5)
    frames = [rng.random((24, 40, 4), dtype=np.float32) for _ in range(9)]
    whole, _ = ctx.sequence_nlm(frames, k=2)
    for start, count in ((0, 3), (3, 3), (6, 3), (0, 1), (4, 5), (8, 1)):
        part, _ = ctx.sequence_nlm(frames, k=2, first=start, count=count)
        assert all(np.array_equal(part[i], whole[start + i]) for i in range(count)), (start, count)


@pytest.mark.gpu
def test_animation_with_a_corrupt_frame_fails_cleanly(tmp_path):
    """One file per worker thread (round 6): a frame that does not decode ends the whole load with the codec's message and exit
    code 1 -- no crash, no hang of the other workers, nothing written."""
    d, frames, _, ext = _make_animation(tmp_path, False, n=6)
    bad = d / f"Animation01_X_0003.{ext}"
    blob = bytearray(bad.read_bytes())
    blob[len(blob) // 2] ^= 0xFF                       # inside the IDAT stream: CRC mismatch
    bad.write_bytes(bytes(blob))
    out = tmp_path / "o"
    out.mkdir()
    r = _run([str(d / f"Animation01_X_0000.{ext}"), "--animation", "--temporal-k", "1", "--outdir", str(out)], tmp_path)
    assert r.returncode == 1, r.stdout + r.stderr
    assert "Animation01_X_0003" in r.stdout + r.stderr or "png:" in r.stdout + r.stderr
    assert not list(out.iterdir())
    # a frame of another size: named, exit code 1
    odd = np.zeros((10, 12, 4), np.uint8)
    mid.save_image(bad, odd)
    r = _run([str(d / f"Animation01_X_0000.{ext}"), "--animation", "--temporal-k", "1", "--outdir", str(out)], tmp_path)
    assert r.returncode == 1 and "size/format differs" in r.stdout + r.stderr, r.stdout + r.stderr

