"""CPU suite, part 1: the oracle itself.

 - orc_cpu_bilateral is pinned against fixtures produced by the reference's own CPU loop
   (tests/golden/ref_cpu_bilateral_*.npz) and, when oracle/_ref is built, against that loop live;
 - the shader restatements are "parity unpinned" by the reference (it has no tests/fixtures and
   its GLSL cannot run here); they are checked from a second direction (float64 NumPy
   restatement of the formulas) and through properties the filters must satisfy.
"""
import os

import numpy as np
import pytest

import np_reference as npr
import oracle
from conftest import GOLDEN, rel_err, synth_hdr, synth_ldr

W0 = lambda h, w: np.zeros((h, w, 8), np.float32)  # noqa: E731


# ---- pinned: the reference's CPU loop ---------------------------------------------------------
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_cpu_bilateral_matches_reference_golden(name):
    g = np.load(os.path.join(GOLDEN, f"ref_cpu_bilateral_{name}.npz"))
    for threads in (1, 4):
        out = oracle.cpu_bilateral(g["img"], int(g["radius"]), 10.0, 0.2, blue_bug=True, threads=threads)
        assert np.array_equal(out, g["out"]), "oracle a7 must be bit-exact with the reference loop"


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("shape,R", [((33, 41), 10), ((21, 25), 10), ((30, 19), 4), ((64, 64), 7)])
def test_cpu_bilateral_matches_reference_live(shape, R):
    rng = np.random.default_rng(R * 100 + shape[0])
    img = rng.random((*shape, 4), dtype=np.float32) * 1.5
    assert np.array_equal(oracle.cpu_bilateral(img, R, 10.0, 0.2, True, 2), oracle.ref_cpu_bilateral(img, R, 2))


@pytest.mark.skipif(not (oracle.have_ref() and oracle.have_ref_as_shipped()), reason="oracle/_ref not built (no /root/reference on this box)")
def test_reference_loop_gives_the_same_bits_at_its_own_compile_flags():
    """The reference ships "-fopenmp -g" with no optimisation level (CMakeLists.txt:31); the checker and the timed baseline use -O2.
    The loop's arithmetic is libm calls in double, stored to float: the two builds agree bit for bit, so the -O2 timing is the
    same computation (bench.py quotes the -O0 rate once beside it)."""
    rng = np.random.default_rng(9)
    img = rng.random((37, 53, 4), dtype=np.float32) * 2.0
    for R, th in ((10, 1), (4, 3)):
        assert np.array_equal(oracle.ref_cpu_bilateral(img, R, th), oracle.ref_cpu_bilateral(img, R, th, as_shipped=True))


def test_cpu_bilateral_reference_quirks():
    """Border stays Pixel{} = 0, alpha forced to 1, blue does not enter the range distance."""
    rng = np.random.default_rng(3)
    img = rng.random((40, 44, 4), dtype=np.float32)
    R = 10
    out = oracle.cpu_bilateral(img, R, 10.0, 0.2, True, 1)
    assert np.all(out[:R] == 0) and np.all(out[:, :R] == 0)
    assert np.all(out[-R + 1:] == 0) and np.all(out[:, -R + 1:] == 0)    # y,x <= dim-R inclusive
    assert np.all(out[R:-R + 1, R:-R + 1, 3] == 1.0)
    img2 = img.copy()
    img2[..., 2] = rng.random((40, 44), dtype=np.float32)                # change blue only
    out2 = oracle.cpu_bilateral(img2, R, 10.0, 0.2, True, 1)
    assert np.array_equal(out[..., :2], out2[..., :2]), "with the blue bug, r/g output ignores blue"
    out3 = oracle.cpu_bilateral(img2, R, 10.0, 0.2, False, 1)
    assert not np.array_equal(out2[..., :2], out3[..., :2])


# ---- regression fixtures of the restatement (unpinned) ------------------------------------------
def test_shader_restatement_regression():
    g = np.load(os.path.join(GOLDEN, "shader_restatement.npz"))
    hdr, ldr = g["hdr"], g["ldr"]
    h, w = hdr.shape[:2]
    assert np.array_equal(oracle.bilateral_texture(hdr, 4, 2.0, 0.2), g["bil_tex_r4"])
    assert np.array_equal(oracle.bilateral_linear(hdr, 4, 2.0, 0.2), g["bil_lin_r4"])
    assert np.array_equal(oracle.bilateral_texture(oracle.unpack_u8(ldr, 0), 8, 2.0, 0.2), g["bil_tex_r8_ldr"])
    W = oracle.bilateral_layers_accum(hdr, g["layer0"], W0(h, w), 4, 2.0, 0.2)
    W = oracle.bilateral_layers_accum(hdr, g["layer1"], W, 4, 2.0, 0.2)
    assert np.array_equal(W, g["layers_W"]) and np.array_equal(oracle.normalize(W), g["layers_out"])
    Wn = oracle.nlm_accum(g["nlm_in_t"], g["nlm_in_n"], W0(h, w), 0.5, (-7, 7), (-3, 3))
    assert np.array_equal(Wn, g["nlm_ref_W"]) and np.array_equal(oracle.normalize(Wn), g["nlm_ref_out"])
    assert np.array_equal(oracle.nlm_accum(g["nlm_in_t"], g["nlm_in_n"], W0(h, w), 0.5, (-10, 11), (-3, 4)), g["nlm_bench_W"])


# ---- second direction: float64 formulas -----------------------------------------------------------
@pytest.mark.parametrize("R", [1, 4, 8])
def test_bilateral_vs_float64(R):
    rng = np.random.default_rng(R)
    img = synth_hdr(rng, 23, 31, 2.0)
    num, den = npr.bilateral_texture(img, R, 2.0, 0.2)
    assert rel_err(oracle.bilateral_texture(img, R, 2.0, 0.2), num / den[..., None]) < 1e-5
    assert rel_err(oracle.bilateral_linear(img, R, 2.0, 0.2), npr.bilateral_linear(img, R, 2.0, 0.2)) < 1e-5


def test_layers_vs_float64():
    rng = np.random.default_rng(5)
    img, lay = synth_hdr(rng, 19, 27), synth_ldr(rng, 19, 27)
    W = oracle.bilateral_layers_accum(img, lay, W0(19, 27), 4, 2.0, 0.2)
    num, den = npr.bilateral_texture(img, 4, 2.0, 0.2, guide=lay.astype(np.float64) / 255.0)
    assert rel_err(W[..., :4], num) < 1e-5 and rel_err(W[..., 4], den) < 1e-5


@pytest.mark.parametrize("search,patch", [((-7, 7), (-3, 3)), ((-10, 11), (-3, 4)), ((-2, 3), (-1, 2))])
def test_nlm_vs_float64(search, patch):
    rng = np.random.default_rng(11)
    t = (synth_hdr(rng, 17, 21) * 0.25).astype(np.float32)
    nb = (t * rng.gamma(16.0, 1 / 16.0, (17, 21, 1))).astype(np.float32)
    W = oracle.nlm_accum(t, nb, W0(17, 21), 0.5, search, patch)
    num, den = npr.nlm_sums(t, nb, 0.5, search, patch)
    assert rel_err(W[..., :4], num) < 2e-5 and rel_err(W[..., 4], den) < 2e-5


@pytest.mark.parametrize("search,patch", [((-7, 7), (-3, 3)), ((-10, 11), (-3, 4)), ((-3, 6), (-2, 3)), ((-4, 5), (0, 1))])
def test_torch_float64_checker_agrees_with_the_oracle_and_the_numpy_restatement(search, patch):
    """tests/f64_checker.py (difference image + box sums, torch float64) is what the GPU suite holds WHOLE 1080p frames
    against (test_gpu_nlm_fullframe.py); here it is held against oracle.c (fp32, shader loops) and against
    np_reference.py (float64, tap loop) on a small two-frame case: three statements of nonlocal.comp:28-63 written
    three different ways must agree -- the float64 pair to float64 rounding, oracle.c to the NLM tolerance."""
    import f64_checker as f64
    rng = np.random.default_rng(12)
    h, w = 19, 26
    t = (synth_hdr(rng, h, w) * 0.25).astype(np.float32)
    nb = (t * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32)
    num, den = f64.nlm_sums(t, [t, nb], 0.5, search, patch)
    num, den = num.cpu().numpy(), den.cpu().numpy()
    n1, d1 = npr.nlm_sums(t, t, 0.5, search, patch)
    n2, d2 = npr.nlm_sums(t, nb, 0.5, search, patch)
    assert rel_err(num, n1 + n2) < 1e-12 and rel_err(den, d1 + d2) < 1e-12
    W = oracle.nlm_accum(t, nb, oracle.nlm_accum(t, t, W0(h, w), 0.5, search, patch), 0.5, search, patch)
    assert rel_err(W[..., :4], num) < 2e-5 and rel_err(W[..., 4], den) < 2e-5
    out = f64.nlm_temporal_output([t, nb], 0, 1, 0.5, search, patch)
    assert rel_err(oracle.normalize(W), out) < 2e-5


@pytest.mark.parametrize("R", [1, 4, 8])
def test_torch_float64_bilateral_checker_agrees_with_the_oracle_and_the_numpy_restatement(R):
    """f64_checker.bilateral_sums (flat-padded shifts, torch float64: what tests/test_gpu_bilateral_fullframe.py holds whole
    1080p frames against) vs np_reference.py (float64) and oracle.c (fp32), both addressings and the layer-guided form."""
    import f64_checker as f64
    rng = np.random.default_rng(20 + R)
    img, lay = synth_hdr(rng, 22, 29, 2.0), synth_ldr(rng, 22, 29)
    num, den = f64.bilateral_sums(img, img, R, 2.0, 0.2)
    n2, d2 = npr.bilateral_texture(img, R, 2.0, 0.2)
    assert rel_err(num.cpu().numpy(), n2) < 1e-12 and rel_err(den.cpu().numpy(), d2) < 1e-12
    assert rel_err(oracle.bilateral_texture(img, R, 2.0, 0.2), (num / den[..., None]).cpu().numpy()) < 1e-5
    num, den = f64.bilateral_sums(img, img, R, 2.0, 0.2, linear=True)
    lin = (num / den[..., None]).cpu().numpy()
    assert rel_err(lin, npr.bilateral_linear(img, R, 2.0, 0.2)) < 1e-12
    assert rel_err(oracle.bilateral_linear(img, R, 2.0, 0.2), lin) < 1e-5
    num, den = f64.bilateral_sums(img, lay.astype(np.float32) / np.float32(255), R, 2.0, 0.2)
    Wl = oracle.bilateral_layers_accum(img, lay, W0(22, 29), R, 2.0, 0.2)
    assert rel_err(Wl[..., :4], num.cpu().numpy()) < 1e-5 and rel_err(Wl[..., 4], den.cpu().numpy()) < 1e-5


# ---- properties ---------------------------------------------------------------------------------
@pytest.mark.parametrize("search,patch", [((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))])
def test_nlm_step_edge_known_answers_derived_from_the_shader_text(search, patch):
    """The oracle's NLM -- the checker of the headline kernel, which nothing in the reference pins -- against known answers worked out BY
    HAND from nonlocal.comp's text for a vertical step edge (tests/np_reference.py::nlm_step_edge_known_answer): weights strictly between 0
    and 1, the plain-sum patch distance, the half-open ranges, the 0.001 bias.  Independent of the oracle's loops and of the float64 checker."""
    from np_reference import nlm_step_edge_known_answer
    h, w, xe = 48, 96, 47
    A, B = np.float32([0.30, 0.50, 0.20, 1.0]), np.float32([0.38, 0.44, 0.26, 1.0])
    img = np.empty((h, w, 4), np.float32)
    img[:, :xe], img[:, xe:] = A, B
    for hp in (0.5, 0.2):
        want = nlm_step_edge_known_answer(w, xe, A, B, hp, search, patch)
        got = oracle.nlm_temporal([img], k=0, hparam=hp, search=search, patch=patch)[0]
        m = 14
        assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (h - 2 * m, w - 2 * m, 4))) < 1e-5      # (fp32 sums of up to 441 terms; SURVEY 8c: 2e-5)
        # the answers are not trivial: weights across the edge are neither 0 nor 1, and the edge pixels move
        assert 1e-3 < abs(want[xe - 1, 0] - A[0]) and abs(want[xe - 1, 0] - A[0]) < abs(B[0] - A[0])
        # the same edge lying HORIZONTALLY (the formula is symmetric in the two axes; the loops are not)
        got_t = oracle.nlm_temporal([np.ascontiguousarray(img.transpose(1, 0, 2))], k=0, hparam=hp, search=search, patch=patch)[0]
        assert rel_err(got_t[m:-m, m:-m], np.broadcast_to(want[m:-m, None, :], (w - 2 * m, h - 2 * m, 4))) < 1e-5


def test_nlm_temporal_step_edge_known_answers():
    """Three step-edge frames of different colours, the middle one filtered over all three (k = 1): the per-frame 0.001 bias and the
    accumulation over neighbour frames (nonlocal.comp:61-62, loop src/main.cpp:1577-1606) against the hand-derived closed form -- for the
    oracle (fp32) and the float64 checker."""
    import f64_checker as f64
    from np_reference import nlm_step_edge_known_answer
    h, w, xe, m = 48, 96, 47, 14
    cols = [(np.float32([0.30, 0.50, 0.20, 1.0]), np.float32([0.38, 0.44, 0.26, 1.0])),
            (np.float32([0.33, 0.47, 0.22, 1.0]), np.float32([0.36, 0.46, 0.21, 1.0])),
            (np.float32([0.27, 0.52, 0.25, 1.0]), np.float32([0.41, 0.40, 0.24, 1.0]))]
    frames = []
    for a, b in cols:
        f = np.empty((h, w, 4), np.float32)
        f[:, :xe], f[:, xe:] = a, b
        frames.append(f)
    for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
        want = nlm_step_edge_known_answer(w, xe, cols[1][0], cols[1][1], 0.5, search, patch, neighbours=cols)
        got = oracle.nlm_temporal(frames, k=1, hparam=0.5, search=search, patch=patch, first=1, count=1)[0]
        assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (h - 2 * m, w - 2 * m, 4))) < 1e-5
        ref = np.asarray(f64.nlm_temporal_output(frames, 1, 1, 0.5, search, patch))
        assert np.abs(ref[m:-m, m:-m] - want[m:-m]).max() < 1e-13


def _column_frames(rng, h, w, n_layers):
    """An image and guide layers whose colours depend on the column only: random walks, so that range weights are neither 0 nor 1."""
    walk = lambda lo, hi, step: np.clip(np.cumsum(rng.normal(0, step, (w, 3)), 0) + rng.uniform(lo, hi, 3), lo, hi)
    img_cols = np.concatenate([walk(0.0, 3.0, 0.15), np.ones((w, 1))], 1).astype(np.float32)
    layer_cols = [np.concatenate([walk(0, 255, 12.0), np.full((w, 1), 255.0)], 1).astype(np.uint8) for _ in range(n_layers)]
    img = np.ascontiguousarray(np.broadcast_to(img_cols, (h, w, 4)))
    layers = [np.ascontiguousarray(np.broadcast_to(lc, (h, w, 4))) for lc in layer_cols]
    return img_cols, layer_cols, img, layers


@pytest.mark.parametrize("R", [4, 8])
def test_layer_guided_bilateral_known_answers_with_layers_that_differ_from_the_image(R):
    """a3 with layer != image (the reference-run fixtures only cover layer == image): the oracle's accumulate + normalize against the closed
    form worked out by hand from bialteral_layers.comp's text for column-only frames (tests/np_reference.py): range weight from the LAYER,
    colour from the IMAGE, UNORM decode, three layers accumulated before the division."""
    from np_reference import bilateral_layers_columns_known_answer
    rng = np.random.default_rng(40 + R)
    h, w = 3 * R + 8, 120
    img_cols, layer_cols, img, layers = _column_frames(rng, h, w, 3)
    want = bilateral_layers_columns_known_answer(img_cols, layer_cols, R, 2.0, 0.2)
    Wb = np.zeros((h, w, 8), np.float32)
    for l in layers:
        Wb = oracle.bilateral_layers_accum(img, l, Wb, R, 2.0, 0.2)
    got = oracle.normalize(Wb)
    assert rel_err(got[R:-R, R:-R], np.broadcast_to(want[R:-R], (h - 2 * R, w - 2 * R, 4))) < 1e-5
    # not trivial: the result differs from the image and from a plain (image-guided) bilateral
    assert np.abs(want[R:-R, :3] - img_cols[R:-R, :3]).max() > 0.05
    assert rel_err(oracle.bilateral_texture(img, R, 2.0, 0.2)[R:-R, R:-R], np.broadcast_to(want[R:-R], (h - 2 * R, w - 2 * R, 4))) > 1e-3


@pytest.mark.parametrize("R", [4, 10])
def test_plain_bilateral_known_answers_general_colours(R):
    """a1 / a2 in the oracle against the hand-derived closed form with guide == image (general colours; the reference-run fixtures are
    blue-constant): column-only and row-only frames, both addressings, interior."""
    from np_reference import bilateral_layers_columns_known_answer
    rng = np.random.default_rng(60 + R)
    h, w = 3 * R + 6, 110
    walk = lambda n: np.clip(np.cumsum(rng.normal(0, 0.06, (n, 3)), 0) + rng.uniform(0.2, 1.5, 3), 0.0, 3.0)
    cols = np.concatenate([walk(w), np.ones((w, 1))], 1).astype(np.float32)
    img = np.ascontiguousarray(np.broadcast_to(cols, (h, w, 4)))
    want = bilateral_layers_columns_known_answer(cols, [cols], R, 2.0, 0.2)
    rows = cols[:h]
    img_t = np.ascontiguousarray(np.broadcast_to(rows[:, None, :], (h, w, 4)))
    want_t = bilateral_layers_columns_known_answer(rows, [rows], R, 2.0, 0.2)
    for f in (oracle.bilateral_texture, oracle.bilateral_linear):
        assert rel_err(f(img, R, 2.0, 0.2)[R:-R, R:-R], np.broadcast_to(want[R:-R], (h - 2 * R, w - 2 * R, 4))) < 5e-6
        assert rel_err(f(img_t, R, 2.0, 0.2)[R:-R, R:-R], np.broadcast_to(want_t[R:-R, None, :], (h - 2 * R, w - 2 * R, 4))) < 5e-6


def test_float64_bilateral_checker_reproduces_the_layer_known_answers_to_rounding():
    """tests/f64_checker.py's bilateral sums (what the whole-frame GPU tests of the bilateral kernels are held against), guide != image, three
    layers accumulated: the hand-derived closed form to 1e-12."""
    import f64_checker as f64
    from np_reference import bilateral_layers_columns_known_answer
    rng = np.random.default_rng(77)
    R, h, w = 6, 30, 90
    img_cols, layer_cols, img, layers = _column_frames(rng, h, w, 3)
    want = bilateral_layers_columns_known_answer(img_cols, layer_cols, R, 2.0, 0.2)
    num, den = 0, 0
    for l in layers:
        n_, d_ = f64.bilateral_sums(img, l.astype(np.float32) / np.float32(255.0), R, 2.0, 0.2)
        num, den = num + n_.cpu().numpy(), den + d_.cpu().numpy()
    got = num / den[..., None]
    assert np.abs(got[R:-R, R:-R] - want[R:-R]).max() < 1e-12


def test_nlm_temporal_k2_known_answers_for_general_column_profiles():
    """configs[4]'s window (+-2 frames) on five frames with INDEPENDENT random-walk column profiles: every weight is a generic value in (0, 1).
    Oracle (fp32) and float64 checker against tests/np_reference.py::nlm_columns_known_answer (the step-edge derivation for any profile)."""
    import f64_checker as f64
    from np_reference import nlm_columns_known_answer
    rng = np.random.default_rng(3)
    h, w, m = 44, 96, 14
    walk = lambda n: np.clip(np.cumsum(rng.normal(0, 0.02, (n, 3)), 0) + rng.uniform(0.2, 0.8, 3), 0, 2)
    fr_cols = [np.concatenate([walk(w), np.ones((w, 1))], 1).astype(np.float32) for _ in range(5)]
    frames = [np.ascontiguousarray(np.broadcast_to(c, (h, w, 4))) for c in fr_cols]
    for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
        want = nlm_columns_known_answer(fr_cols[2], 0.5, search, patch, neighbour_cols=fr_cols)
        got = oracle.nlm_temporal(frames, k=2, hparam=0.5, search=search, patch=patch, first=2, count=1)[0]
        assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (h - 2 * m, w - 2 * m, 4))) < 1e-5
        ref = np.asarray(f64.nlm_temporal_output(frames, 2, 2, 0.5, search, patch))
        assert np.abs(ref[m:-m, m:-m] - want[m:-m]).max() < 1e-12
        assert np.abs(want[m:-m, :3] - fr_cols[2][m:-m, :3]).max() > 0.02


def test_nlm_known_answers_for_frames_that_vary_in_both_axes():
    """colour(x, y) = f(x) + g(y): the patch distance needs 1-D sums only (tests/np_reference.py::nlm_additive_known_answer, worked out by
    hand from nonlocal.comp's text), yet both box-sum axes, every search offset and the cross term carry generic values.  Temporal k = 1 over
    three such frames: float64 checker to 1e-13, oracle to 1e-5."""
    import f64_checker as f64
    from conftest import additive_frames
    from np_reference import nlm_additive_known_answer
    rng = np.random.default_rng(9)
    h, w = 50, 70
    frs = additive_frames(rng, h, w, 3)
    for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
        m = max(-search[0], search[1] - 1) + max(-patch[0], patch[1] - 1)
        want = nlm_additive_known_answer(frs[1][0], frs[1][1], 0.5, search, patch, neighbours=[(a, b) for a, b, _ in frs])
        got = oracle.nlm_temporal([x[2] for x in frs], k=1, hparam=0.5, search=search, patch=patch, first=1, count=1)[0]
        assert rel_err(got[m:-m, m:-m], want[m:-m, m:-m]) < 1e-5
        ref = np.asarray(f64.nlm_temporal_output([x[2] for x in frs], 1, 1, 0.5, search, patch))
        assert np.abs(ref[m:-m, m:-m] - want[m:-m, m:-m]).max() < 1e-13
        assert np.abs(want[m:-m, m:-m, :3] - frs[1][2][m:-m, m:-m, :3]).max() > 0.03


def test_float64_checker_reproduces_the_step_edge_known_answers_to_rounding():
    """tests/f64_checker.py -- the independent float64 evaluation every whole-frame GPU test of the NLM kernels is held against -- gives the
    hand-derived closed form to 1e-13, both edge orientations, both tuned windows: shader text -> closed form -> checker -> kernels."""
    import f64_checker as f64
    from np_reference import nlm_step_edge_known_answer
    h, w, xe, m = 48, 96, 47, 14
    A, B = np.float32([0.30, 0.50, 0.20, 1.0]), np.float32([0.38, 0.44, 0.26, 1.0])
    img = np.empty((h, w, 4), np.float32)
    img[:, :xe], img[:, xe:] = A, B
    for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
        want = nlm_step_edge_known_answer(w, xe, A, B, 0.5, search, patch)
        ref = np.asarray(f64.nlm_temporal_output([img], 0, 0, 0.5, search, patch))
        assert np.abs(ref[m:-m, m:-m] - want[m:-m]).max() < 1e-13
        ref_t = np.asarray(f64.nlm_temporal_output([np.ascontiguousarray(img.transpose(1, 0, 2))], 0, 0, 0.5, search, patch))
        assert np.abs(ref_t[m:-m, m:-m] - want[m:-m, None, :]).max() < 1e-13


def test_constant_image_is_invariant_in_the_interior():
    img = np.tile(np.array([0.3, 0.6, 0.9, 1.0], np.float32), (30, 34, 1))
    R = 4
    for f in (oracle.bilateral_texture, oracle.bilateral_linear):
        out = f(img, R, 2.0, 0.2)
        assert rel_err(out[R:-R, R:-R], img[R:-R, R:-R]) < 1e-6
    corner = oracle.bilateral_texture(img, R, 2.0, 0.2)[0, 0]
    assert np.all(corner < img[0, 0]), "zero texels outside the image darken the border (reference behaviour)"
    W = oracle.nlm_accum(img, img, W0(30, 34), 0.5, (-2, 3), (-1, 2))
    o = oracle.normalize(W)[8:-8, 8:-8]
    assert rel_err(o * (25.001 / 25.0), img[8:-8, 8:-8]) < 1e-6, "0.001 bias on the norm (nonlocal.comp:32)"


def test_texture_and_linear_agree_in_the_interior_and_differ_at_row_ends():
    rng = np.random.default_rng(2)
    img = synth_hdr(rng, 26, 33)
    R = 4
    a, b = oracle.bilateral_texture(img, R, 2.0, 0.2), oracle.bilateral_linear(img, R, 2.0, 0.2)
    assert rel_err(a[:, R:-R], b[:, R:-R]) < 1e-6      # same taps, transposed loop order only
    assert rel_err(a[:, :R], b[:, :R]) > 1e-3          # row wrap-around vs zero border


def test_accumulation_is_linear_over_frames_and_layers():
    rng = np.random.default_rng(8)
    t = rng.random((14, 18, 4), dtype=np.float32)
    n1, n2 = rng.random((14, 18, 4), dtype=np.float32), rng.random((14, 18, 4), dtype=np.float32)
    z = W0(14, 18)
    w1 = oracle.nlm_accum(t, n1, z, 0.5, (-2, 3), (-1, 2))
    w2 = oracle.nlm_accum(t, n2, z, 0.5, (-2, 3), (-1, 2))
    w12 = oracle.nlm_accum(t, n2, w1, 0.5, (-2, 3), (-1, 2))
    assert np.array_equal(w12[..., :5], (w1 + w2)[..., :5]), "W += is a plain fp32 add of per-dispatch sums"
    assert np.allclose(w12[..., 4].min(), 0.002, atol=1e-3) or w12[..., 4].min() > 0.002


def test_normalize_sentinel_and_division():
    W = W0(2, 3)
    W[0, 0] = [1, 2, 3, 4, 2, 0, 0, 0]
    W[0, 1] = [5, 5, 5, 5, 0, 0, 0, 0]          # normWeight == 0 -> magenta (normalize.comp:36-38)
    W[1, 2] = [1, 1, 1, 1, 3, 9, 9, 9]          # padding is ignored
    out = oracle.normalize(W)
    assert np.array_equal(out[0, 0], np.float32([0.5, 1, 1.5, 2]))
    assert np.array_equal(out[0, 1], np.float32([1, 0, 1, 1]))
    assert np.array_equal(out[1, 2], np.float32([1, 1, 1, 1]) / np.float32(3))
    assert np.array_equal(out[1, 0], np.float32([1, 0, 1, 1]))


def test_u8_paths_exhaustive():
    u8 = np.arange(256, dtype=np.uint8)
    unorm, cpu = oracle.unpack_u8(u8, 0), oracle.unpack_u8(u8, 1)
    assert np.array_equal(unorm, (u8.astype(np.float32) / np.float32(255)))
    assert np.array_equal(cpu, u8.astype(np.float32) * (np.float32(1) / np.float32(255)))
    assert unorm[255] == 1.0 and unorm[0] == 0.0
    assert (unorm != cpu).sum() > 0, "the two decode flavours differ by 1 ulp for some codes"
    # truncating pack (unsigned char)(255*v): both decode flavours survive the round trip for all
    # 256 codes (fl(c/255)*255 rounds back to >= c), which is why the reference's PNG path is lossless
    assert np.array_equal(oracle.pack_u8(unorm), u8) and np.array_equal(oracle.pack_u8(cpu), u8)
    just_below = np.nextafter(unorm[1:], np.float32(0))
    assert np.array_equal(oracle.pack_u8(just_below), u8[1:] - 1), "truncation, not rounding"
    edge = np.float32([0.0, 1.0, 0.999999, -0.0, -0.5, -2.0, 1.5, np.nan, 254.999 / 255, 1 / 255, 0.00392])
    assert list(oracle.pack_u8(edge)) == [0, 255, 254, 0, 0, 0, 255, 0, 254, 1, 0]


def test_threaded_nlm_restatement_is_the_same_arithmetic():
    """bench.py's cpu_baseline spreads the shader invocations over OpenMP threads: results must not move."""
    rng = np.random.default_rng(31)
    t, nb = synth_hdr(rng, 23, 41), synth_hdr(rng, 23, 41)
    W = rng.random((23, 41, 8)).astype(np.float32)
    one = oracle.nlm_accum(t, nb, W, 0.5, (-10, 11), (-3, 4))
    for th in (2, 5):
        assert np.array_equal(oracle.nlm_accum(t, nb, W, 0.5, (-10, 11), (-3, 4), threads=th), one)


def test_three_instruction_unorm8_decode_equals_the_ieee_quotient_for_all_codes():
    """csrc/common.hpp unorm8(): q = c*(1/255); q += (c - 255 q)*(1/255) with two fmas.  In exact rational arithmetic
    with round-to-nearest-even to 24 bits this is the correctly rounded c/255 (what UNORM decode and the oracle's
    c/255.0f give) for every one of the 256 codes; the plain product alone is not (126 codes differ by one ulp)."""
    from fractions import Fraction
    import math

    def rnd32(x):
        if x == 0:
            return Fraction(0)
        s, x = (1 if x > 0 else -1), abs(x)
        e = math.floor(math.log2(x))
        while Fraction(2) ** e > x:
            e -= 1
        while Fraction(2) ** (e + 1) <= x:
            e += 1
        ulp = Fraction(2) ** (e - 23)
        q = x / ulp
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return s * n * ulp

    k = rnd32(Fraction(1, 255))
    assert float(k) == float(np.float32(1.0) / np.float32(255.0))
    product_only_wrong = 0
    for c in range(256):
        want = rnd32(Fraction(c, 255))
        assert float(want) == float(np.float32(c) / np.float32(255.0))
        q0 = rnd32(Fraction(c) * k)
        product_only_wrong += q0 != want
        r = rnd32(Fraction(c) - 255 * q0)          # fma: exact product-sum, one rounding
        assert rnd32(r * k + q0) == want, c
    assert product_only_wrong == 126
    assert np.array_equal(oracle.unpack_u8(np.arange(256, dtype=np.uint8), 0),
                          (np.arange(256, dtype=np.float32) / np.float32(255.0)))
