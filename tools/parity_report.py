"""The numbers behind DESIGN.md section 6: largest deviation of the NLM kernels from the float64 torch checker over WHOLE 1080p
frames (the cases of tests/test_gpu_nlm_fullframe.py), and of the 1x1-patch NLM from the reference-anchored bilateral
(tests/test_gpu_reference_anchor.py).  The tests assert < 2e-5; this prints what the margin is.  GPU box: python tools/parity_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import f64_checker as f64
import image_denoising_filter_amd as mid
from conftest import rel_err, synth_hdr

H, W = 1080, 1920
ctx = mid.Context(0)
rng = np.random.default_rng(2)
scene = synth_hdr(rng, H, W + 16, 6.0) * np.float32(0.25)
frames = []
for i in range(5):
    f = scene[:, 2 * i:2 * i + W] * rng.gamma(16.0, 1 / 16.0, (H, W, 1))
    f[..., 3] = 1.0
    frames.append(np.ascontiguousarray(f, dtype=np.float32))
for name, cfg in (("21x21/7x7", dict(search=(-10, 11), patch=(-3, 4))), ("[-7,7)/[-3,3)", dict(search=(-7, 7), patch=(-3, 3)))):
    got = ctx.nlm_temporal([frames[0]], k=0, **cfg)[0]
    ref = f64.nlm_temporal_output([frames[0]], 0, 0, 0.5, cfg["search"], cfg["patch"])
    print(f"NLM {name:14s} k=0, every pixel of 1920x1080 vs float64: max |d|/max(1,|ref|) = {rel_err(got, ref):.2e}")
got = ctx.nlm_temporal(frames, k=2, first=2, count=1, search=(-10, 11), patch=(-3, 4))[0]
ref = f64.nlm_temporal_output(frames, 2, 2, 0.5, (-10, 11), (-3, 4))
print(f"NLM 21x21/7x7      k=2 (5 frames), every pixel vs float64:             max = {rel_err(got, ref):.2e}")
for R in (10, 8):
    r2 = np.random.default_rng(77 + R)
    img = (synth_hdr(r2, H, W, 6.0) * 0.25 * r2.gamma(16.0, 1 / 16.0, (H, W, 1))).astype(np.float32)
    img[..., 3] = r2.random((H, W), dtype=np.float32)
    Wn = ctx.nlm_accum(img, img, np.zeros((H, W, 8), np.float32), 0.5, (-R, R + 1), (0, 1))
    bil = ctx.bilateral(img, R, 1e6, 0.5 / np.sqrt(2.0), "texture").astype(np.float64)
    sw = Wn[..., 4].astype(np.float64) - 0.001
    print(f"NLM 1x1 patch, {2 * R + 1}x{2 * R + 1} search vs the reference-anchored bilateral r={R}, whole noisy frame incl. borders: max = {rel_err(Wn[..., :4].astype(np.float64) / sw[..., None], bil):.2e}")
frame = synth_hdr(np.random.default_rng(2), H, W, 6.0)
for layout in ("texture", "linear"):
    num, den = f64.bilateral_sums(frame, frame, 8, 2.0, 0.2, linear=(layout == "linear"))
    print(f"bilateral r=8 {layout:7s}, every pixel of 1920x1080 vs float64: max = {rel_err(ctx.bilateral(frame, 8, 2.0, 0.2, layout), (num / den[..., None]).cpu().numpy()):.2e}")
num, den = f64.bilateral_sums(frame, frame, 20, 2.0, 0.2)
print(f"bilateral r=20 texture (TEXEL_WINDOW as shipped), every pixel vs float64:  max = {rel_err(ctx.bilateral(frame, 20, 2.0, 0.2, 'texture'), (num / den[..., None]).cpu().numpy()):.2e}")
from conftest import synth_ldr
r3 = np.random.default_rng(10)
layers = [synth_ldr(r3, H, W) for _ in range(4)]
num = den = None
for lay in layers:
    n_, d_ = f64.bilateral_sums(frame, lay.astype(np.float32) / np.float32(255.0), 8, 2.0, 0.2)
    num, den = (n_, d_) if num is None else (num + n_, den + d_)
print(f"layer-guided bilateral r=8, 4 RGBA8 layers fused, every pixel vs float64:  max = {rel_err(ctx.bilateral_layers(frame, layers, 8, 2.0, 0.2), (num / den[..., None]).cpu().numpy()):.2e}")
