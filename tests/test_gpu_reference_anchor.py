"""GPU suite: the kernels held DIRECTLY against output of the reference itself -- no oracle in between.

The only code of the reference that can run here is its CPU bilateral loop (src/main.cpp:1827-1864, compiled into
oracle/_ref in the build container; tests/golden/ref_blue_const_*.npz hold its inputs and outputs).  On an image whose
blue channel is constant the loop's typo `texColor.b - texColor.b` (:1850) is also the true blue difference, so what
the loop computes on the interior IS the formula of bialteral.comp / bialteral_linear.comp / bialteral_layers.comp with
sigma_s=10, sigma_c=0.2.  That anchors:
    a2 (linear bilateral)   interior + the wrapped column w-R     direct
    a1 (texture bilateral)  interior                              direct
    a3 (layer-guided)       layer == image, one layer             direct (accumulate + normalize, and the fused form)
    a5 (normalize)          inside the a3 chain                   direct
    a6 (u8 decode)          inside the RGBA8 cases                direct (UNORM c/255 vs the CPU path's c*(1/255): <= 1 ulp in)
a4 (NLM) cannot be anchored this way: the reference has no CPU NLM.
Tolerance: 1e-5 * max(1, |ref|) (the loop evaluates its weights in double, the kernels in fp32).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _fix(name):
    g = np.load(os.path.join(GOLDEN, f"ref_blue_const_{name}.npz"))
    return g["img"], g["img_u8"], int(g["radius"]), g["out"]


@pytest.mark.parametrize("name", ["d", "e", "f", "g"])
def test_a1_a2_bilateral_equal_the_reference_run(ctx, name):
    img, u8, R, out = _fix(name)
    h, w = img.shape[:2]
    lin = ctx.bilateral(img, R, 10.0, 0.2, "linear")
    tex = ctx.bilateral(img, R, 10.0, 0.2, "texture")
    # linear: column w-R included (its taps wrap into the next row exactly like the loop's flat index, :1840-1842);
    # rows h-R-1.. read zero pixels past the image, where typo and true blue difference part ways
    assert rel_err(lin[R:h - R - 1, R:w - R + 1, :3], out[R:h - R - 1, R:w - R + 1, :3]) < TOL
    assert rel_err(tex[R:h - R, R:w - R, :3], out[R:h - R, R:w - R, :3]) < TOL
    assert rel_err(tex[R:h - R, R:w - R, 3], out[R:h - R, R:w - R, 3]) < TOL          # alpha == 1.0 (:1863)
    if u8.size:
        # the RGBA8 path: UNORM decode in the kernel (src/texture.cpp:16) vs the CPU path's c*(1/255) (:1804-1807)
        tex8 = ctx.bilateral(u8, R, 10.0, 0.2, "texture")
        lin8 = ctx.bilateral(u8, R, 10.0, 0.2, "linear")
        assert rel_err(tex8[R:h - R, R:w - R], out[R:h - R, R:w - R]) < TOL
        assert rel_err(lin8[R:h - R - 1, R:w - R + 1, :3], out[R:h - R - 1, R:w - R + 1, :3]) < TOL


@pytest.mark.parametrize("name", ["e", "g"])
def test_a3_a5_layer_guided_with_layer_equal_image_equals_the_reference_run(ctx, name):
    """bialteral_layers.comp takes the range distance from the layer and the colour from the image (:29,47-55): with
    layer == image (RGBA8, one layer) it is the plain bilateral, so accumulate -> normalize must reproduce the
    reference loop's output on the interior; the fused entry point gives the bits of that sequence."""
    _, u8, R, out = _fix(name)
    h, w = u8.shape[:2]
    W = ctx.bilateral_layers_accum(u8, u8, np.zeros((h, w, 8), np.float32), R, 10.0, 0.2)
    seq = ctx.normalize(W)
    assert rel_err(seq[R:h - R, R:w - R], out[R:h - R, R:w - R]) < TOL
    fused = ctx.bilateral_layers(u8, [u8], R, 10.0, 0.2)
    assert np.array_equal(fused, seq)
    # and with the float image as input, RGBA8 guide: colour path RGBA32F, range path UNORM-decoded
    img, _, _, _ = _fix(name)
    fl = ctx.bilateral_layers(img, [u8], R, 10.0, 0.2)
    assert rel_err(fl[R:h - R, R:w - R], out[R:h - R, R:w - R]) < TOL


# ---- a4: the nearest thing to an anchor the reference allows ------------------------------------------------------
def _ramp(h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return np.stack([0.2 + 0.004 * xx + 0.001 * yy, 0.9 - 0.002 * xx + 0.003 * yy, 0.5 + 0.0015 * xx - 0.0025 * yy,
                     np.ones_like(xx)], -1).astype(np.float32)


@pytest.mark.parametrize("patch", [(-3, 4), (-3, 3), (-1, 2)])
def test_a4_nlm_on_a_linear_ramp_is_the_anchored_bilateral(ctx, patch):
    """The reference has no CPU NLM, so a4 cannot be held against reference-run output.  What CAN be done: on an image
    that is linear in x and y every patch tap sees the same colour difference, d(p,s) = P^2 * |I(p) - I(p+s)|^2
    (P^2 = number of patch taps, nonlocal.comp:42-52: a plain sum over the half-open patch), so nonlocal.comp's weight
    exp(-d/h^2) is bialteral.comp's range weight with sigma_c = h / sqrt(2 P^2) and no spatial term -- and the GPU
    bilateral IS anchored to the reference run (tests above).  This pins, through a1: the tap count of the half-open
    patch range, the exponent, the search window, the 4-channel weighted sum, and the 0.001 the shader adds to the norm
    (nonlocal.comp:32)."""
    h, w, R, hp = 60, 76, 6, 0.5
    img = _ramp(h, w)
    P2 = (patch[1] - patch[0]) ** 2
    Wn = ctx.nlm_accum(img, img, np.zeros((h, w, 8), np.float32), hp, (-R, R + 1), patch)
    bil = ctx.bilateral(img, R, 1e6, hp / np.sqrt(2.0 * P2), "texture")
    m = R + max(-patch[0], patch[1])                        # away from the zero texels beyond the border
    sw = Wn[m:-m, m:-m, 4].astype(np.float64) - 0.001       # sum of weights without the shader's bias
    got = Wn[m:-m, m:-m, :4].astype(np.float64) / sw[..., None]
    assert rel_err(got, bil[m:-m, m:-m]) < 2e-5
    # and the fused form: normalize divides by (0.001 + sum w)
    out = ctx.nlm_temporal([img], k=0, hparam=hp, search=(-R, R + 1), patch=patch)[0]
    assert rel_err(out[m:-m, m:-m], bil[m:-m, m:-m].astype(np.float64) * (sw / (sw + 0.001))[..., None]) < 2e-5


@pytest.mark.parametrize("R", [10, 8])
def test_a4_nlm_with_a_1x1_patch_on_a_noisy_1080p_frame_is_the_anchored_bilateral(ctx, R):
    """The ramp identity above needs a linear image; this one holds on ANY image: with a 1x1 patch (patch range [0,1))
    nonlocal.comp's distance is the plain colour difference of the two texels, d(p,s) = |I(p) - I(p+s)|^2_rgb
    (nonlocal.comp:42-52 with one tap), so its weight exp(-d/h^2) (:55) is bialteral.comp's range weight
    exp(-d / (2 sigma_c^2)) (bialteral.comp:60-65) with sigma_c = h / sqrt(2), and sigma_s -> infinity removes the
    spatial term (exp(-r^2 / 2e12) rounds to 1 in fp32 for r <= 15).  Both shaders sum the 4-channel colour of the
    (2R+1)^2 texels of the window (nonlocal.comp:36-38,56 with the closed range [-R, R+1); bialteral.comp:51-72), and
    both give an out-of-image texel the value vec4(0) AND its weight -- so the identity holds on the WHOLE frame,
    borders included, on noisy data where the weights are all different.  The bilateral kernel is anchored to output
    of the reference's own loop (tests above), so this pins through a1: the walk over the search window (every offset
    visited once, none transposed: the image is not symmetric), the zero-texel border policy, the exponent, the
    4-channel weighted sum and the 0.001 norm bias (nonlocal.comp:32) -- on a full-size frame."""
    from conftest import synth_hdr
    H, W, hp = 1080, 1920, 0.5
    rng = np.random.default_rng(77 + R)
    img = (synth_hdr(rng, H, W, 6.0) * 0.25 * rng.gamma(16.0, 1 / 16.0, (H, W, 1))).astype(np.float32)
    img[..., 3] = rng.random((H, W), dtype=np.float32)               # alpha is carried like colour by both shaders
    Wn = ctx.nlm_accum(img, img, np.zeros((H, W, 8), np.float32), hp, (-R, R + 1), (0, 1))
    bil = ctx.bilateral(img, R, 1e6, hp / np.sqrt(2.0), "texture").astype(np.float64)
    sw = Wn[..., 4].astype(np.float64) - 0.001                        # >= 1: the zero offset has weight exp(0)
    assert sw.min() > 0.999
    got = Wn[..., :4].astype(np.float64) / sw[..., None]
    assert rel_err(got, bil) < 2e-5
    # the fused form divides by (0.001 + sum w): mid_nlm_temporal's output is the bilateral's, scaled by sw / (sw + 0.001)
    out = ctx.nlm_temporal([img], k=0, hparam=hp, search=(-R, R + 1), patch=(0, 1))[0]
    assert rel_err(out, bil * (sw / (sw + 0.001))[..., None]) < 2e-5
    # not vacuous: the filter moved the frame, and the borders are darker than the interior's mean ratio
    assert rel_err(out, img) > 1e-2
    assert (out[0, :, 0] / np.maximum(img[0, :, 0], 1e-6)).mean() < (out[540, :, 0] / np.maximum(img[540, :, 0], 1e-6)).mean()
