"""GPU suite: the drop-in CLI at the reference's OWN defaults on 1920x1080 inputs -- all six GPU modes of main()
(src/main.cpp:1952-1973), a PNG animation directory and an EXR one, every pixel of every output against the float64 torch
evaluation of tests/f64_checker.py.

Why this exists (LABNOTES R5.1): tests/test_cli.py drives the same six modes on 40x56 frames, where no copy is
larger than 36 KB.  Here each mode decodes 8-33 MB images, hands them to mid_memcpy_h2d / mid_nlm_multiframe and reads a 33 MB
result back -- the product's large-copy path, through page-locked decode buffers (mid_image_load_pinned) as shipped; a second
run with --pageable-host keeps every host buffer in ordinary memory, so the library's own bounce buffers
(csrc/hostcopy.cpp) carry the same frames, and must produce the same files.

Defaults exercised: bilateral TEXEL_WINDOW 20 / sigma 2.0, 0.2 (bialteral.comp:5, src/main.cpp:806), NLM [-7,7) x [-3,3), h 0.5
(nonlocal.comp:5-6, src/main.cpp:870), frame list = target + every sibling (src/main.cpp:1390-1393), layers = the
RenderElements files that carry the target's 4-digit id (src/main.cpp:1343-1378)."""
import os
import subprocess

import numpy as np
import pytest

import f64_checker as f64
import image_denoising_filter_amd as mid
from conftest import ROOT, rel_err, synth_hdr, synth_ldr

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "image_denoising_filter_amd", "mi_denoise")
H, W, N_FRAMES, TARGET = 1080, 1920, 3, 1
NAMES = ["output-nonlinear-bialteral", "output-nonlinear-bialteral-layers", "output-linear-bialteral",
         "output-nonlinear-nlm", "output-nonlinear-nlm-multiframe", "output-nonlinear-nlm-multiframe-overlap"]


def _make(root, hdr):
    rng = np.random.default_rng(31 + hdr)
    d = root / "Anim"
    (d / "RenderElements").mkdir(parents=True)
    ext = "exr" if hdr else "png"
    base = synth_hdr(rng, H, W, 2.0) * 0.25
    frames, layers = [], []
    for i in range(N_FRAMES):
        f = (np.roll(base, 2 * i, axis=1) * rng.gamma(16.0, 1 / 16.0, (H, W, 1))).astype(np.float32)
        if not hdr:
            f = (np.clip(f, 0, 1) * 255).astype(np.uint8)
            f[..., 3] = 255
        frames.append(f)
        mid.save_image(d / f"Animation01_X_{i:04d}.{ext}", f)
    for name in ("albedo", "normal"):
        layers.append(synth_ldr(rng, H, W))
        mid.save_image(d / "RenderElements" / f"{name}_{TARGET:04d}.png", layers[-1])
    return d, frames, layers, ext


def _expected(frames, layers, hdr):
    """float64 outputs of the six modes at the CLI's defaults."""
    f32 = [f if hdr else f.astype(np.float32) / np.float32(255.0) for f in frames]       # UNORM decode c/255, src/texture.cpp:16
    t = f32[TARGET]
    out = {}
    for name, linear in ((NAMES[0], False), (NAMES[2], True)):
        num, den = f64.bilateral_sums(t, t, 20, 2.0, 0.2, linear=linear)
        out[name] = (num / den[..., None]).cpu().numpy()
    num = den = None
    for lay in layers:
        n_, d_ = f64.bilateral_sums(t, lay.astype(np.float32) / np.float32(255.0), 20, 2.0, 0.2)
        num, den = (n_, d_) if num is None else (num + n_, den + d_)
    out[NAMES[1]] = (num / den[..., None]).cpu().numpy()
    num, den = f64.nlm_sums(t, [t], 0.5, (-7, 7), (-3, 3))
    out[NAMES[3]] = (num / den[..., None]).cpu().numpy()
    num, den = f64.nlm_sums(t, [t] + f32, 0.5, (-7, 7), (-3, 3))                         # target first, then every sibling
    out[NAMES[4]] = out[NAMES[5]] = (num / den[..., None]).cpu().numpy()
    return out


def _pack_u8(x):
    """(unsigned char)(255.0f * v), src/main.cpp:97-103 -- on the float32 rounding of the float64 expectation."""
    v = np.float32(255.0) * x.astype(np.float32)
    return np.clip(np.trunc(v), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("hdr", [False, True])
def test_six_reference_modes_at_1080p_every_pixel(tmp_path, hdr):
    d, frames, layers, ext = _make(tmp_path, hdr)
    want = _expected(frames, layers, hdr)
    outs = {}
    for pageable in (False, True):
        out = tmp_path / ("out_pageable" if pageable else "out_pinned")
        out.mkdir()
        r = subprocess.run([CLI, str(d / f"Animation01_X_{TARGET:04d}.{ext}"), "--gpu-only", "--outdir", str(out)]
                           + (["--pageable-host"] if pageable else []), cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert r.stdout.count("Running on GPU") == 6 and r.stdout.count("transfer time: ") == 6
        assert ("host buffers: pageable" in r.stdout) == pageable
        assert sorted(os.listdir(out)) == sorted(f"{n}.{ext}" for n in NAMES)
        outs[pageable] = {n: mid.load_image(out / f"{n}.{ext}") for n in NAMES}
    for n in NAMES:
        got = outs[False][n]
        assert np.array_equal(got, outs[True][n]), f"{n}: pinned and pageable host buffers gave different files"
        assert got.shape == (H, W, 4)
        if hdr:
            assert rel_err(got, want[n]) < 2e-5, n
        else:
            diff = np.abs(got.astype(np.int16) - _pack_u8(want[n]).astype(np.int16))
            # truncating u8 pack: a 1e-6 float difference moves a value across an integer boundary now and then
            assert diff.max() <= 1 and (diff[..., :3] != 0).mean() < 1e-3, (n, int(diff.max()), float((diff != 0).mean()))
    # not vacuous: the six modes are different filters
    assert not np.array_equal(outs[False][NAMES[0]], outs[False][NAMES[3]])
    assert not np.array_equal(outs[False][NAMES[3]], outs[False][NAMES[4]])


@pytest.mark.parametrize("hdr", [True, False])
def test_animation_mode_at_1080p_pinned_and_pageable_frames(tmp_path, hdr):
    """--animation streams every frame through mid_sequence_nlm_range[_u8]: 33 MB (EXR) / 8 MB (PNG) per copy each way.  Default:
    frames decoded into page-locked memory; `--pinned-mb 0`: every frame in ordinary memory, carried by the library's bounce
    buffers.  Same files both ways, every pixel against the float64 temporal evaluation (PNG outputs: the u8 read-back
    conversion runs on the device, compared within one LSB of the truncation boundary)."""
    d, frames, _, ext = _make(tmp_path, hdr)
    f32 = [f if hdr else f.astype(np.float32) / np.float32(255.0) for f in frames]
    outs = {}
    for label, extra in (("pinned", []), ("pageable", ["--pinned-mb", "0"])):
        out = tmp_path / label
        out.mkdir()
        r = subprocess.run([CLI, str(d / f"Animation01_X_0000.{ext}"), "--animation", "--temporal-k", "1", "--outdir", str(out)] + extra,
                           cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert ("beyond the page-locked budget" in r.stdout) == (label == "pageable")
        outs[label] = [mid.load_image(out / f"output-animation-Animation01_X_{i:04d}.{ext}") for i in range(N_FRAMES)]
    for i in range(N_FRAMES):
        assert np.array_equal(outs["pinned"][i], outs["pageable"][i]), i
        want = f64.nlm_temporal_output(f32, i, 1, 0.5, (-7, 7), (-3, 3))
        if hdr:
            assert rel_err(outs["pinned"][i], want) < 2e-5, i
        else:
            diff = np.abs(outs["pinned"][i].astype(np.int16) - _pack_u8(want).astype(np.int16))
            assert diff.max() <= 1 and (diff[..., :3] != 0).mean() < 1e-3, (i, int(diff.max()))
