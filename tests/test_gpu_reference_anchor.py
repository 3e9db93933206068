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
