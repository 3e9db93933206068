"""GPU suite at BASELINE.json's full size (1920x1080): the oracle cannot run a whole frame in
seconds, so parity is carried by (i) oracle runs on windows cut around tile seams, frame corners
and random interior spots of the full-size GPU result, and (ii) size-independent properties."""
import numpy as np
import pytest

import oracle
from conftest import rel_err, synth_hdr

pytestmark = pytest.mark.gpu
H, W = 1080, 1920


@pytest.fixture(scope="module")
def frame():
    rng = np.random.default_rng(2)
    return synth_hdr(rng, H, W, 6.0)


def _windows(rng, halo, size=24):
    """(y0, x0) of windows: the four corners, edges, tile seams (x=58,59,64; y=16,64) and random spots."""
    pts = [(0, 0), (0, W - size), (H - size, 0), (H - size, W - size), (0, 900), (H - size, 1000),
           (500, 0), (600, W - size), (64 - 12, 58 - 12), (16 - 8, 64 - 12), (128 - 12, 116 - 12), (1024, 1856)]
    pts += [(int(rng.integers(0, H - size)), int(rng.integers(0, W - size))) for _ in range(4)]
    return pts


def _crop(img, y0, x0, size, halo):
    """Window + halo, zero beyond the image (the texture policy), plus the offsets of the window inside it."""
    ya, yb, xa, xb = y0 - halo, y0 + size + halo, x0 - halo, x0 + size + halo
    out = np.zeros((yb - ya, xb - xa, 4), np.float32)
    sy, sx = slice(max(ya, 0), min(yb, H)), slice(max(xa, 0), min(xb, W))
    out[sy.start - ya:sy.stop - ya, sx.start - xa:sx.stop - xa] = img[sy, sx]
    return out


@pytest.mark.parametrize("R", [8, 20])
def test_bilateral_fullsize_windows(ctx, frame, R):
    rng = np.random.default_rng(R)
    out = ctx.bilateral(frame, R, 2.0, 0.2, "texture")
    size = 24
    for y0, x0 in _windows(rng, R, size):
        c = _crop(frame, y0, x0, size, R)
        ref = oracle.bilateral_texture(c, R, 2.0, 0.2)[R:R + size, R:R + size]
        assert rel_err(out[y0:y0 + size, x0:x0 + size], ref) < 1e-5, (y0, x0)


def test_bilateral_linear_equals_texture_away_from_row_ends(ctx, frame):
    a = ctx.bilateral(frame, 8, 2.0, 0.2, "texture")
    b = ctx.bilateral(frame, 8, 2.0, 0.2, "linear")
    assert np.array_equal(a[:, 8:-8], b[:, 8:-8]), "same tile code, same taps: identical bits in the interior"
    assert not np.array_equal(a[:, :8], b[:, :8])
    # the linear variant's first columns see the previous row's tail: check a few against the oracle
    c = frame[100:140].copy()
    ref = oracle.bilateral_linear(c, 8, 2.0, 0.2)
    assert rel_err(b[100 + 10:100 + 30, :24], ref[10:30, :24]) < 1e-5
    assert rel_err(b[100 + 10:100 + 30, -24:], ref[10:30, -24:]) < 1e-5


@pytest.mark.parametrize("cfg", [dict(search=(-10, 11), patch=(-3, 4)), dict(search=(-7, 7), patch=(-3, 3))])
def test_nlm_fullsize_windows(ctx, frame, cfg):
    rng = np.random.default_rng(7)
    t = (frame * 0.25).astype(np.float32)
    nb = (t * rng.gamma(16.0, 1 / 16.0, (H, W, 1))).astype(np.float32)
    out = ctx.nlm_temporal([t, nb], k=1, first=0, count=1, **cfg)[0]
    halo = max(-cfg["search"][0], cfg["search"][1]) + max(-cfg["patch"][0], cfg["patch"][1])
    size = 16
    for y0, x0 in _windows(rng, halo, size)[:12]:
        ct, cn = _crop(t, y0, x0, size, halo), _crop(nb, y0, x0, size, halo)
        Wz = np.zeros((*ct.shape[:2], 8), np.float32)
        Wz = oracle.nlm_accum(ct, ct, Wz, 0.5, **cfg)
        Wz = oracle.nlm_accum(ct, cn, Wz, 0.5, **cfg)
        ref = oracle.normalize(Wz)[halo:halo + size, halo:halo + size]
        assert rel_err(out[y0:y0 + size, x0:x0 + size], ref) < 2e-5, (y0, x0)


def test_nlm_batch_equals_single_launches_and_is_deterministic(ctx, frame):
    rng = np.random.default_rng(9)
    fr = [(frame * 0.25 * rng.gamma(16.0, 1 / 16.0, (H, W, 1))).astype(np.float32) for _ in range(3)]
    cfg = dict(search=(-10, 11), patch=(-3, 4))
    batch = ctx.nlm_temporal(fr, k=0, **cfg)
    again = ctx.nlm_temporal(fr, k=0, **cfg)
    for i in range(3):
        single = ctx.nlm_temporal([fr[i]], k=0, **cfg)[0]
        assert np.array_equal(batch[i], single) and np.array_equal(batch[i], again[i])


@pytest.mark.parametrize("cfg", [dict(search=(-10, 11), patch=(-3, 4)), dict(search=(-7, 7), patch=(-3, 3))])
@pytest.mark.parametrize("ldr", [False, True])
def test_half_shape_of_a_small_launchs_last_round_is_bit_identical(ctx, frame, cfg, ldr):
    """One 1080p frame is 1156 workgroups = 2 full rounds + 132: the 132 run in the HALF shape (eight waves per tile, half a
    strip each, csrc/nlm_strip.hpp), fused and unfused alike; a 3-frame launch (6 rounds + 396) runs entirely in the standard
    shape.  Same bits, for both tuned windows and both texel formats -- and the tail covers the bottom-right tiles, so
    the ragged last tile row (1080 = 33 x 32 + 24) and column are in it."""
    rng = np.random.default_rng(21)
    if ldr:
        fr = [np.clip(frame[..., :4] * 0.12 * rng.gamma(16.0, 1 / 16.0, (H, W, 1)), 0, 1) for _ in range(3)]
        fr = [(f * 255).astype(np.uint8) for f in fr]
        for f in fr:
            f[..., 3] = 255
    else:
        fr = [(frame * 0.25 * rng.gamma(16.0, 1 / 16.0, (H, W, 1))).astype(np.float32) for _ in range(3)]
    batch = ctx.nlm_temporal(fr, k=0, **cfg)
    for i in (0, 2):
        single = ctx.nlm_temporal([fr[i]], k=0, **cfg)[0]
        assert np.array_equal(batch[i], single), i
    unfused = ctx.normalize(ctx.nlm_accum(fr[1], fr[1], np.zeros((H, W, 8), np.float32), 0.5, **cfg))
    assert np.array_equal(batch[1], unfused)
    # temporal launches never use the shape; a one-output temporal launch over the same frames still agrees with the sequence
    t1 = ctx.nlm_temporal(fr, k=1, first=1, count=1, **cfg)[0]
    Wt = np.zeros((H, W, 8), np.float32)
    for f in fr:
        Wt = ctx.nlm_accum(fr[1], f, Wt, 0.5, **cfg)
    assert np.array_equal(t1, ctx.normalize(Wt))


@pytest.mark.parametrize("h", [1083, 1086])
def test_half_shape_on_a_ragged_last_tile_row(ctx, frame, h):
    """Heights that are not a multiple of 8: in the last tile row a strip is cut after 3 rows (1083 % 8 = 3: the HALF
    shape's upper-half wave writes 3 rows, its lower-half wave none) or after 6 (1086: the lower-half wave writes 2 of
    its 4).  Still 34 x 34 tiles = 1156 workgroups = 2 full rounds + 132, so the tail rule fires exactly as at 1080 rows,
    and the XCD remap puts the first half of the bottom tile row into the standard shape and the second half into the
    HALF shape (on a 256-CU device).  Same bits as the 3-frame launch (no tail) and as accumulate + normalize, and every
    pixel within tolerance of the float64 checker."""
    import f64_checker as f64
    rng = np.random.default_rng(h)
    big = np.concatenate([frame, frame[:16]], 0)[:h]
    fr = [(big * 0.25 * rng.gamma(16.0, 1 / 16.0, (h, W, 1))).astype(np.float32) for _ in range(3)]
    for cfg in (dict(search=(-10, 11), patch=(-3, 4)), dict(search=(-7, 7), patch=(-3, 3))):
        batch = ctx.nlm_temporal(fr, k=0, **cfg)
        single = ctx.nlm_temporal([fr[2]], k=0, **cfg)[0]
        assert np.array_equal(batch[2], single)
        unfused = ctx.normalize(ctx.nlm_accum(fr[2], fr[2], np.zeros((h, W, 8), np.float32), 0.5, **cfg))
        assert np.array_equal(single, unfused)
        ref = f64.nlm_temporal_output([fr[2]], 0, 0, 0.5, cfg["search"], cfg["patch"])
        assert rel_err(single, ref) < 2e-5
        assert rel_err(single[-8:], ref[-8:]) < 2e-5 and single[-8:].min() > 0


def test_constant_frame_properties(ctx):
    img = np.tile(np.float32([0.25, 0.5, 2.0, 1.0]), (H, W, 1))
    out = ctx.bilateral(img, 8, 2.0, 0.2)
    assert rel_err(out[8:-8, 8:-8], img[8:-8, 8:-8]) < 1e-6
    o = ctx.nlm_temporal([img], k=0, search=(-10, 11), patch=(-3, 4))[0]
    # (the strip kernel carries sqrt(log2 e)/h in the colours, so 441 equal addends are no longer powers of two:
    #  ordinary fp32 summation error, inside the NLM tolerance)
    assert rel_err(o[13:-13, 13:-13] * (441.001 / 441.0), img[13:-13, 13:-13]) < 2e-5
    assert np.all(o[0, 0, :3] < img[0, 0, :3])          # zero texels beyond the border take weight


def test_u8_round_trip_fullsize(ctx):
    rng = np.random.default_rng(12)
    u8 = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    f = ctx.unpack_u8(u8, 0)
    assert np.array_equal(ctx.pack_u8(f), u8)
    assert int(f.view(np.uint32).sum(dtype=np.uint64)) == int(oracle.unpack_u8(u8, 0).view(np.uint32).sum(dtype=np.uint64))


def test_bilateral_scale_covariance_fullsize(ctx, frame):
    """bilateral(a*I; sigma_c*a) == a*bilateral(I; sigma_c) for a power of two a: every product and difference scales
    exactly, the exponent argument is unchanged -> identical bits up to the final scale."""
    a = ctx.bilateral(frame, 8, 2.0, 0.2)
    b = ctx.bilateral(frame * np.float32(4.0), 8, 2.0, 0.8)
    assert rel_err(b, a * np.float32(4.0)) < 2e-6
    # alpha is carried like colour (bialteral.comp:67,72): scaled too
    assert rel_err(b[..., 3], a[..., 3] * 4.0) < 2e-6


def test_nlm_limits_of_the_filtering_parameter_fullsize(ctx, frame):
    """h -> 0: only the zero offset keeps weight 1 (d = 0), so out = I / 1.001 (the 0.001 norm bias, nonlocal.comp:32);
    h -> large: every weight -> 1, so out = (21x21 box mean of I) * 441/441.001 in the interior."""
    t = (frame * 0.25).astype(np.float32)
    cfg = dict(search=(-10, 11), patch=(-3, 4))
    tiny = ctx.nlm_temporal([t], k=0, hparam=1e-3, **cfg)[0]
    assert rel_err(tiny * np.float32(1.001), t) < 2e-6
    big = ctx.nlm_temporal([t], k=0, hparam=1e4, **cfg)[0]
    # box mean over offsets [-10, 11) in both axes, via a float64 summed-area table
    sat = np.zeros((H + 1, W + 1, 4), np.float64)
    sat[1:, 1:] = t.astype(np.float64).cumsum(0).cumsum(1)
    ys, xs = np.arange(10, H - 10), np.arange(10, W - 10)
    box = (sat[ys[:, None] + 11, xs[None, :] + 11] - sat[ys[:, None] - 10, xs[None, :] + 11]
           - sat[ys[:, None] + 11, xs[None, :] - 10] + sat[ys[:, None] - 10, xs[None, :] - 10]) / 441.001
    assert rel_err(big[10:-10, 10:-10], box) < 2e-5


@pytest.mark.parametrize("search,patch", [((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))])
def test_nlm_step_edge_known_answers_fullsize(ctx, search, patch):
    """The kernels against known answers derived by hand from nonlocal.comp's text (tests/np_reference.py::nlm_step_edge_known_answer): a
    1920x1080 frame with a vertical step edge that is NOT on a tile seam, every interior pixel held against the closed form -- a check of the
    headline kernel that involves neither the oracle nor the float64 checker."""
    from np_reference import nlm_step_edge_known_answer
    xe = 1003
    A, B = np.float32([0.30, 0.50, 0.20, 1.0]), np.float32([0.38, 0.44, 0.26, 1.0])
    img = np.empty((H, W, 4), np.float32)
    img[:, :xe], img[:, xe:] = A, B
    for hp in (0.5, 0.2):
        want = nlm_step_edge_known_answer(W, xe, A, B, hp, search, patch)
        got = ctx.nlm_temporal([img], k=0, hparam=hp, search=search, patch=patch)[0]
        m = 14
        assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (H - 2 * m, W - 2 * m, 4))) < 2e-5
    # the same edge lying HORIZONTALLY, between two strips of a tile (row 517): the vertical box sums and the strip seams
    ye = 517
    img[:ye], img[ye:] = A, B
    want = nlm_step_edge_known_answer(H, ye, A, B, 0.5, search, patch)
    got = ctx.nlm_temporal([img], k=0, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m, None, :], (H - 2 * m, W - 2 * m, 4))) < 2e-5
    # three step-edge frames of different colours, the middle one filtered over all three (temporal k = 1): the per-frame 0.001 bias and
    # the accumulation over neighbour frames (nonlocal.comp:61-62) in the fused multi-frame kernel
    cols = [(A, B), (np.float32([0.33, 0.47, 0.22, 1.0]), np.float32([0.36, 0.46, 0.21, 1.0])),
            (np.float32([0.27, 0.52, 0.25, 1.0]), np.float32([0.41, 0.40, 0.24, 1.0]))]
    frames = []
    for a, b in cols:
        f = np.empty((H, W, 4), np.float32)
        f[:, :xe], f[:, xe:] = a, b
        frames.append(f)
    want = nlm_step_edge_known_answer(W, xe, cols[1][0], cols[1][1], 0.5, search, patch, neighbours=cols)
    got = ctx.nlm_temporal(frames, k=1, first=1, count=1, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (H - 2 * m, W - 2 * m, 4))) < 2e-5


@pytest.mark.parametrize("search,patch", [((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))])
def test_nlm_temporal_k2_known_answers_for_general_profiles_fullsize(ctx, search, patch):
    """configs[4]'s window (+-2 frames) at 1920x1080 on five frames with independent random-walk profiles -- column-only, then row-only -- so that
    every weight is a generic value in (0, 1): the fused multi-frame kernel on every interior pixel against the hand-derived closed form
    (tests/np_reference.py::nlm_columns_known_answer); the single-frame launch likewise."""
    from np_reference import nlm_columns_known_answer
    rng = np.random.default_rng(41)
    m = 14
    walk = lambda n: np.clip(np.cumsum(rng.normal(0, 0.02, (n, 3)), 0) + rng.uniform(0.2, 0.8, 3), 0, 2)
    fr_cols = [np.concatenate([walk(W), np.ones((W, 1))], 1).astype(np.float32) for _ in range(5)]
    frames = [np.ascontiguousarray(np.broadcast_to(c, (H, W, 4))) for c in fr_cols]
    want = nlm_columns_known_answer(fr_cols[2], 0.5, search, patch, neighbour_cols=fr_cols)
    got = ctx.nlm_temporal(frames, k=2, first=2, count=1, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (H - 2 * m, W - 2 * m, 4))) < 2e-5
    assert np.abs(want[m:-m, :3] - fr_cols[2][m:-m, :3]).max() > 0.02
    want1 = nlm_columns_known_answer(fr_cols[0], 0.5, search, patch)
    got1 = ctx.nlm_temporal([frames[0]], k=0, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got1[m:-m, m:-m], np.broadcast_to(want1[m:-m], (H - 2 * m, W - 2 * m, 4))) < 2e-5
    rows = [c[:H] for c in fr_cols]
    frames_t = [np.ascontiguousarray(np.broadcast_to(r[:, None, :], (H, W, 4))) for r in rows]
    want_t = nlm_columns_known_answer(rows[2], 0.5, search, patch, neighbour_cols=rows)
    got_t = ctx.nlm_temporal(frames_t, k=2, first=2, count=1, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got_t[m:-m, m:-m], np.broadcast_to(want_t[m:-m, None, :], (H - 2 * m, W - 2 * m, 4))) < 2e-5
    # RGBA8 frames (the reference's default PNG path: UNORM texels c / 255 in fp32, src/texture.cpp:16), h = 0.2 so that 8-bit steps matter
    u8_cols = [np.concatenate([np.clip(np.cumsum(rng.normal(0, 1.5, (W, 3)), 0) + rng.uniform(60, 200, 3), 0, 255), np.full((W, 1), 255.0)], 1).astype(np.uint8)
               for _ in range(3)]
    frames_u8 = [np.ascontiguousarray(np.broadcast_to(c, (H, W, 4))) for c in u8_cols]
    dec = [c.astype(np.float32) / np.float32(255.0) for c in u8_cols]
    want8 = nlm_columns_known_answer(dec[1], 0.2, search, patch, neighbour_cols=dec)
    got8 = ctx.nlm_temporal(frames_u8, k=1, first=1, count=1, hparam=0.2, search=search, patch=patch)[0]
    assert rel_err(got8[m:-m, m:-m], np.broadcast_to(want8[m:-m], (H - 2 * m, W - 2 * m, 4))) < 2e-5
    assert np.abs(want8[m:-m, :3] - dec[1][m:-m, :3]).max() > 0.005


def test_translation_equivariance_away_from_the_borders(ctx, frame):
    """Shifting the frame shifts the result: tiles, strips and wave seams land on different pixels, so this catches any
    dependence on the position inside a tile beyond rounding."""
    t = (frame * 0.25).astype(np.float32)
    dy, dx = 5, 37
    sh = np.roll(t, (dy, dx), axis=(0, 1))
    cfg = dict(search=(-10, 11), patch=(-3, 4))
    a = ctx.nlm_temporal([t], k=0, **cfg)[0]
    b = ctx.nlm_temporal([sh], k=0, **cfg)[0]
    m = 13 + max(dy, dx)
    assert rel_err(b[m:-m, m:-m], np.roll(a, (dy, dx), axis=(0, 1))[m:-m, m:-m]) < 2e-6
    ba = ctx.bilateral(t, 8, 2.0, 0.2)
    bb = ctx.bilateral(sh, 8, 2.0, 0.2)
    assert np.array_equal(bb[m:-m, m:-m], np.roll(ba, (dy, dx), axis=(0, 1))[m:-m, m:-m]), "bilateral taps are summed in tap order: bit-identical"
