"""GPU suite: WHOLE 1920x1080 frames of the headline kernel (mid_nlm_temporal, through the C-ABI) against an independent
float64 evaluation -- every pixel, tolerance 2e-5 * max(1, |ref|).

The reference holds nothing that pins nonlocal.comp (no CPU NLM, no fixture, no test: DESIGN.md section 6), and
oracle.c needs minutes per 1080p frame, so until round 4 the full-size evidence was 12-16 oracle windows of 16x16 px
(0.2 % of a frame) plus GPU-vs-GPU bit-identity chains.  Here the checker is tests/f64_checker.py: the difference-image
/ box-sum form of the same formula in torch float64, executed on this box's device, sharing no code with oracle.c or
with the kernels, and held against oracle.c on a small frame inside this file (and against np_reference.py in the CPU
suite).  What is covered: the 21x21/7x7 benchmark window and the shipped [-7,7)/[-3,3) window at k=0 (one launch =
2 full rounds of workgroups in the standard shape + the last round in the HALF shape), and one k=2 output of a
five-frame sequence (the temporal kernel: five tile fills, per-frame 0.001 bias, totals kept apart from per-frame sums).
"""
import numpy as np
import pytest

import f64_checker as f64
import oracle
from conftest import rel_err, synth_hdr

pytestmark = pytest.mark.gpu
H, W = 1080, 1920
NLM_TOL = 2e-5
BENCH = dict(search=(-10, 11), patch=(-3, 4))
REFERENCE = dict(search=(-7, 7), patch=(-3, 3))


@pytest.fixture(scope="module")
def frames():
    """Five frames of the bench's kind: one HDR scene (radiance up to ~1.6 after the 0.25 scale), independent
    multiplicative Monte-Carlo-like noise per frame, a 2 px pan per frame."""
    rng = np.random.default_rng(2)
    scene = synth_hdr(rng, H, W + 16, 6.0) * np.float32(0.25)
    out = []
    for i in range(5):
        f = scene[:, 2 * i:2 * i + W] * rng.gamma(16.0, 1 / 16.0, (H, W, 1))
        f[..., 3] = 1.0
        out.append(np.ascontiguousarray(f, dtype=np.float32))
    return out


def test_the_checker_runs_on_the_device_and_agrees_with_the_oracle_on_a_small_frame(ctx):
    """Same code path as the full-frame tests (torch float64 on cuda:0), at a size oracle.c finishes in a second."""
    assert f64.device().type == "cuda", "the full-frame checker is meant to run on the GPU box's device"
    rng = np.random.default_rng(5)
    h, w = 45, 83
    fr = [(synth_hdr(rng, h, w) * 0.25 * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32) for _ in range(3)]
    for cfg in (BENCH, REFERENCE):
        ref = f64.nlm_temporal_output(fr, 1, 1, 0.5, cfg["search"], cfg["patch"])
        assert rel_err(oracle.nlm_temporal(fr, k=1, first=1, count=1, **cfg)[0], ref) < NLM_TOL
        assert rel_err(ctx.nlm_temporal(fr, k=1, first=1, count=1, **cfg)[0], ref) < NLM_TOL


@pytest.mark.parametrize("name,cfg", [("bench", BENCH), ("reference", REFERENCE)])
def test_nlm_single_frame_every_pixel_of_1080p(ctx, frames, name, cfg):
    got = ctx.nlm_temporal([frames[0]], k=0, **cfg)[0]
    ref = f64.nlm_temporal_output([frames[0]], 0, 0, 0.5, cfg["search"], cfg["patch"])
    assert got.shape == ref.shape == (H, W, 4)
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    worst = np.unravel_index(int(np.argmax(err)), err.shape)
    assert err.max() < NLM_TOL, (name, float(err.max()), worst)
    # not a vacuous comparison: the filter moved the frame, and the border rows saw zero texels
    assert rel_err(got, frames[0]) > 1e-2
    # the unfused dispatch sequence (mid_nlm_accum + mid_normalize) is the same bits, so it is covered too
    Wz = ctx.nlm_accum(frames[0], frames[0], np.zeros((H, W, 8), np.float32), 0.5, **cfg)
    assert np.array_equal(ctx.normalize(Wz), got)


def test_nlm_temporal_k2_every_pixel_of_one_1080p_output(ctx, frames):
    got = ctx.nlm_temporal(frames, k=2, first=2, count=1, **BENCH)[0]
    ref = f64.nlm_temporal_output(frames, 2, 2, 0.5, BENCH["search"], BENCH["patch"])
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    worst = np.unravel_index(int(np.argmax(err)), err.shape)
    assert err.max() < NLM_TOL, (float(err.max()), worst)
    # a clipped window at the sequence start (frames 0..2 only) through the same launch path
    got0 = ctx.nlm_temporal(frames, k=2, first=0, count=1, **BENCH)[0]
    ref0 = f64.nlm_temporal_output(frames, 0, 2, 0.5, BENCH["search"], BENCH["patch"])
    assert rel_err(got0, ref0) < NLM_TOL
