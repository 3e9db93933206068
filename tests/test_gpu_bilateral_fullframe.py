"""GPU suite: WHOLE 1920x1080 frames of the bilateral kernels (mid_bilateral, mid_bilateral_layers, through the C-ABI) against the
float64 torch evaluation of tests/f64_checker.py -- every pixel, tolerance 1e-5 * max(1, |ref|).

BASELINE configs[1] (r = 8, linear vs texture addressing) and configs[3] (4 RGBA8 guide layers, r = 8) at their stated size.
tests/test_gpu_fullsize.py / test_gpu_configs.py hold the same launches against oracle.c on windows (0.4 % of a frame) because
oracle.c needs minutes per frame; here nothing is sampled.  The checker shares no code with oracle.c or the kernels and is held
against both the NumPy float64 restatement and oracle.c on a small frame in the CPU suite (tests/test_oracle.py)."""
import numpy as np
import pytest

import f64_checker as f64
from conftest import rel_err, synth_hdr, synth_ldr

pytestmark = pytest.mark.gpu
H, W = 1080, 1920
TOL = 1e-5


@pytest.fixture(scope="module")
def frame():
    rng = np.random.default_rng(2)
    return synth_hdr(rng, H, W, 6.0)


@pytest.mark.parametrize("layout", ["texture", "linear"])
def test_bilateral_r8_every_pixel_of_1080p(ctx, frame, layout):
    got = ctx.bilateral(frame, 8, 2.0, 0.2, layout)
    num, den = f64.bilateral_sums(frame, frame, 8, 2.0, 0.2, linear=(layout == "linear"))
    ref = (num / den[..., None]).cpu().numpy()
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() < TOL, (layout, float(err.max()), np.unravel_index(int(np.argmax(err)), err.shape))
    assert rel_err(got, frame) > 1e-2                                # the filter did something
    if layout == "linear":                                           # and the two addressings differ where they should: at the row ends
        tex = ctx.bilateral(frame, 8, 2.0, 0.2, "texture")
        assert np.array_equal(tex[:, 8:-8], got[:, 8:-8]) and not np.array_equal(tex[:, :8], got[:, :8])


def test_bilateral_r20_shipped_window_every_pixel_of_1080p(ctx, frame):
    """TEXEL_WINDOW 20 as the reference ships it (bialteral.comp:5): 41 x 41 taps."""
    got = ctx.bilateral(frame, 20, 2.0, 0.2, "texture")
    num, den = f64.bilateral_sums(frame, frame, 20, 2.0, 0.2)
    assert rel_err(got, (num / den[..., None]).cpu().numpy()) < TOL


def test_layer_guided_bilateral_four_layers_every_pixel_of_1080p(ctx, frame):
    """configs[3]: range distance from each RGBA8 layer (UNORM-decoded), colour from the float image, sums over the layers in the
    WeightInfo accumulator, then normalize.comp; fused kernel == accumulate x 4 + normalize is pinned elsewhere, bit for bit."""
    rng = np.random.default_rng(10)
    layers = [synth_ldr(rng, H, W) for _ in range(4)]
    got = ctx.bilateral_layers(frame, layers, 8, 2.0, 0.2)
    num = den = None
    for lay in layers:
        n_, d_ = f64.bilateral_sums(frame, lay.astype(np.float32) / np.float32(255.0), 8, 2.0, 0.2)
        num, den = (n_, d_) if num is None else (num + n_, den + d_)
    ref = (num / den[..., None]).cpu().numpy()
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() < TOL, (float(err.max()), np.unravel_index(int(np.argmax(err)), err.shape))


def test_nlm_rgba8_input_every_pixel_of_1080p(ctx):
    """The LDR (.png) path of configs[2]: RGBA8 frames, UNORM-decoded while the tile is filled."""
    rng = np.random.default_rng(11)
    u8 = synth_ldr(rng, H, W)
    got = ctx.nlm_temporal([u8], k=0, search=(-10, 11), patch=(-3, 4))[0]
    ref = f64.nlm_temporal_output([u8.astype(np.float32) / np.float32(255.0)], 0, 0, 0.5, (-10, 11), (-3, 4))
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("layout", ["texture", "linear"])
@pytest.mark.parametrize("R", [8, 20])
def test_plain_bilateral_known_answers_on_column_only_frames(ctx, layout, R):
    """a1 / a2 against known answers worked out by hand from bialteral.comp's text (tests/np_reference.py: for frames whose colours depend on
    the column only the 2-D window collapses to one axis; guide == image is the plain bilateral): general colours -- the reference-run
    fixtures are blue-constant -- on every interior pixel of 1080p, both addressings, neither the oracle nor the float64 checker involved.
    Row-only frames exercise the other axis (and, in the linear addressing, taps that stay inside their row)."""
    from np_reference import bilateral_layers_columns_known_answer
    rng = np.random.default_rng(500 + R)
    walk = lambda n: np.clip(np.cumsum(rng.normal(0, 0.06, (n, 3)), 0) + rng.uniform(0.2, 1.5, 3), 0.0, 3.0)
    cols = np.concatenate([walk(W), np.ones((W, 1))], 1).astype(np.float32)
    img = np.ascontiguousarray(np.broadcast_to(cols, (H, W, 4)))
    want = bilateral_layers_columns_known_answer(cols, [cols], R, 2.0, 0.2)
    got = ctx.bilateral(img, R, 2.0, 0.2, layout)
    assert rel_err(got[R:-R, R:-R], np.broadcast_to(want[R:-R], (H - 2 * R, W - 2 * R, 4))) < 1e-5
    assert np.abs(want[R:-R, :3] - cols[R:-R, :3]).max() > 0.01           # the filter does something
    rows = cols[:H]
    img_t = np.ascontiguousarray(np.broadcast_to(rows[:, None, :], (H, W, 4)))
    want_t = bilateral_layers_columns_known_answer(rows, [rows], R, 2.0, 0.2)
    got_t = ctx.bilateral(img_t, R, 2.0, 0.2, layout)
    assert rel_err(got_t[R:-R, R:-R], np.broadcast_to(want_t[R:-R, None, :], (H - 2 * R, W - 2 * R, 4))) < 1e-5
