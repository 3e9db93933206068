"""GPU suite: the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

Tolerances: integer pack/unpack and normalize bit-exact; bilateral
|d| <= 1e-5*max(1,|ref|); NLM (re-associated box sums) <= 2e-5*max(1,|ref|).
The oracle's shader restatements are "parity unpinned" (see oracle/oracle.h): these tests
establish GPU == restatement, and the restatement is tied to the reference text line by line.
"""
import os

import numpy as np
import pytest

import oracle
import image_denoising_filter_amd as mid
from conftest import GOLDEN, rel_err, synth_hdr, synth_ldr

pytestmark = pytest.mark.gpu

BIL_TOL, NLM_TOL = 1e-5, 2e-5
Z = lambda h, w: np.zeros((h, w, 8), np.float32)  # noqa: E731


def test_native_library_is_the_thing_under_test(ctx):
    assert "gfx950" in ctx.name
    assert os.path.basename(mid.LIB_PATH).startswith("libmi_denoise")


# ---- a1 / a2 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["texture", "linear"])
@pytest.mark.parametrize("R", [1, 3, 4, 6, 8, 10, 13, 20, 24])  # 4/8/10/20 tuned tiles, others the run-time-radius kernel
def test_bilateral_hdr(ctx, layout, R):
    rng = np.random.default_rng(R)
    h, w = (70, 131) if R < 20 else (75, 140)                  # not multiples of the tile, > 1 tile each way
    img = synth_hdr(rng, h, w)
    orc = oracle.bilateral_texture if layout == "texture" else oracle.bilateral_linear
    assert rel_err(ctx.bilateral(img, R, 2.0, 0.2, layout), orc(img, R, 2.0, 0.2)) < BIL_TOL


@pytest.mark.parametrize("layout", ["texture", "linear"])
def test_bilateral_ldr_input_is_unorm_decoded(ctx, layout):
    rng = np.random.default_rng(1)
    img8 = synth_ldr(rng, 61, 97)
    orc = oracle.bilateral_texture if layout == "texture" else oracle.bilateral_linear
    ref = orc(oracle.unpack_u8(img8, 0), 4, 10.0, 0.2)
    assert rel_err(ctx.bilateral(img8, 4, 10.0, 0.2, layout), ref) < BIL_TOL


@pytest.mark.parametrize("shape", [(1, 1), (1, 40), (40, 1), (3, 5), (16, 64), (17, 65)])
def test_bilateral_tiny_and_ragged_frames(ctx, shape):
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.random((*shape, 4), dtype=np.float32) * 3
    for layout, orc in (("texture", oracle.bilateral_texture), ("linear", oracle.bilateral_linear)):
        assert rel_err(ctx.bilateral(img, 8, 2.0, 0.2, layout), orc(img, 8, 2.0, 0.2)) < BIL_TOL


def test_bilateral_sigma_sweep(ctx):
    rng = np.random.default_rng(4)
    img = synth_hdr(rng, 40, 70)
    for ss, sc in ((0.7, 0.05), (2.0, 0.2), (10.0, 0.2), (5.0, 3.0)):
        assert rel_err(ctx.bilateral(img, 4, ss, sc), oracle.bilateral_texture(img, 4, ss, sc)) < BIL_TOL


# ---- a3 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R", [4, 8, 6, 16, 22])               # 22: two tiles exceed LDS -> per-pixel fallback
def test_layers_accumulate_and_fused(ctx, R):
    rng = np.random.default_rng(20 + R)
    h, w = 50, 90
    img = synth_hdr(rng, h, w)
    layers = [synth_ldr(rng, h, w) for _ in range(3)]
    Wg, Wo = Z(h, w), Z(h, w)
    for l in layers:
        Wg = ctx.bilateral_layers_accum(img, l, Wg, R)
        Wo = oracle.bilateral_layers_accum(img, l, Wo, R)
    assert rel_err(Wg[..., :5], Wo[..., :5]) < BIL_TOL * 4     # sums of up to (2R+1)^2 terms, three layers
    fused = ctx.bilateral_layers(img, layers, R)
    assert rel_err(fused, oracle.normalize(Wo)) < BIL_TOL
    assert np.array_equal(fused, ctx.normalize(Wg)), "fused == per-layer accumulate + normalize, bit for bit"


def test_layers_ldr_input_and_no_layers(ctx):
    rng = np.random.default_rng(31)
    img8, lay = synth_ldr(rng, 33, 47), synth_ldr(rng, 33, 47)
    ref = oracle.normalize(oracle.bilateral_layers_accum(oracle.unpack_u8(img8, 0), lay, Z(33, 47), 4))
    assert rel_err(ctx.bilateral_layers(img8, [lay], 4), ref) < BIL_TOL
    # zero layers: the weight buffer is never touched -> normWeight == 0 -> magenta everywhere
    assert np.array_equal(ctx.bilateral_layers(img8, [], 4), np.tile(np.float32([1, 0, 1, 1]), (33, 47, 1)))


# ---- a4 -------------------------------------------------------------------------------------------
NLM_CFGS = {"ref": dict(search=(-7, 7), patch=(-3, 3)), "bench": dict(search=(-10, 11), patch=(-3, 4)),
            "generic": dict(search=(-3, 4), patch=(-1, 2)),           # run-time search range, 3x3 patch
            "rt7": dict(search=(-6, 9), patch=(-3, 4)),               # run-time (asymmetric) search, 7x7 patch
            "rt5": dict(search=(-12, 13), patch=(-2, 3)),             # 25x25 search, 5x5 patch
            "rt4": dict(search=(-4, 5), patch=(-2, 2)),               # 4x4 patch ([-P,P) at P=2) on the strip kernel
            "rt2": dict(search=(-4, 5), patch=(-1, 1)),               # 2x2 patch ([-P,P) at P=1) on the strip kernel
            "rt1": dict(search=(-6, 7), patch=(0, 1)),                # pixel-wise weights (1x1 patch) on the strip kernel
            "rt10": dict(search=(-3, 4), patch=(-5, 5)), "rt11": dict(search=(-5, 6), patch=(-5, 6)),   # 10x10 .. 13x13 patches:
            "rt12": dict(search=(-2, 3), patch=(-6, 6)), "rt13": dict(search=(-4, 5), patch=(-6, 7)),   # strips of four rows
            "rt16": dict(search=(-2, 3), patch=(-8, 8)),              # the largest patch the ABI takes
            "naive": dict(search=(-2, 3), patch=(-1, 3))}             # lopsided 4x4 patch: no strip instantiation -> one-thread-per-pixel fallback


def _nlm_f64(t, nb, W0, hparam, search, patch):
    """W0 + one dispatch's sums from the float64 NumPy restatement (tests/np_reference.py): [h, w, 5] float64."""
    import np_reference as npr
    num, den = npr.nlm_sums(t, nb, hparam, search, patch)
    return np.concatenate([num, den[..., None]], -1) + np.asarray(W0, np.float64)[..., :5]


def _nlm_pair(rng, h, w, scale=0.25):
    t = (synth_hdr(rng, h, w) * scale).astype(np.float32)
    nb = (t * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32)
    return t, nb


@pytest.mark.parametrize("cfg", ["ref", "bench", "generic", "rt7", "rt5", "rt4", "rt2", "rt1", "rt10", "rt11", "rt12", "rt13", "rt16", "naive"])
def test_nlm_accum(ctx, cfg):
    rng = np.random.default_rng(40)
    h, w = 71, 125                                   # > 1 tile in x (58/59 px) and y (64 px), ragged
    t, nb = _nlm_pair(rng, h, w)
    W0 = rng.random((h, w, 8), dtype=np.float32)     # the dispatch ADDS to whatever W holds
    Wg, Wo = ctx.nlm_accum(t, nb, W0, 0.5, **NLM_CFGS[cfg]), oracle.nlm_accum(t, nb, W0, 0.5, **NLM_CFGS[cfg])
    if cfg == "rt5":
        # 625 offsets x 5x5 patch: oracle.c (fp32, the shader's loop order) and the kernel (fp32, block sums) are two
        # fp32 evaluations of sums of 625 terms and can sit on opposite sides of the exact value, so each is held
        # against the float64 restatement at the SAME tolerance instead of against the other at a wider one.
        Wo = _nlm_f64(t, nb, W0, 0.5, **NLM_CFGS[cfg])
        assert rel_err(oracle.nlm_accum(t, nb, W0, 0.5, **NLM_CFGS[cfg])[..., :5], Wo) < NLM_TOL
    assert rel_err(Wg[..., :5], Wo[..., :5]) < NLM_TOL
    assert np.array_equal(Wg[..., 5:], W0[..., 5:]), "std430 padding is never written"


@pytest.mark.parametrize("cfg", ["ref", "bench"])
def test_nlm_hdr_range_and_h_sweep(ctx, cfg):
    rng = np.random.default_rng(41)
    t, nb = _nlm_pair(rng, 40, 66, scale=1.0)        # radiance up to ~8: most weights underflow
    for hp in (0.5, 0.1, 2.0):
        out = ctx.normalize(ctx.nlm_accum(t, nb, Z(40, 66), hp, **NLM_CFGS[cfg]))
        ref = oracle.normalize(oracle.nlm_accum(t, nb, Z(40, 66), hp, **NLM_CFGS[cfg]))
        assert rel_err(out, ref) < NLM_TOL


@pytest.mark.parametrize("shape", [(1, 1), (2, 70), (70, 2), (9, 9), (64, 58), (65, 59)])
def test_nlm_tiny_and_ragged_frames(ctx, shape):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    t = rng.random((*shape, 4), dtype=np.float32)
    nb = np.clip(t + 0.03 * rng.standard_normal((*shape, 4)), 0, 1).astype(np.float32)
    for cfg in ("ref", "bench"):
        Wg = ctx.nlm_accum(t, nb, Z(*shape), 0.5, **NLM_CFGS[cfg])
        assert rel_err(Wg[..., :5], oracle.nlm_accum(t, nb, Z(*shape), 0.5, **NLM_CFGS[cfg])[..., :5]) < NLM_TOL


def test_nlm_ldr_input(ctx):
    rng = np.random.default_rng(43)
    a, b = synth_ldr(rng, 37, 64), synth_ldr(rng, 37, 64)
    ref = oracle.nlm_accum(oracle.unpack_u8(a, 0), oracle.unpack_u8(b, 0), Z(37, 64), 0.5)
    assert rel_err(ctx.nlm_accum(a, b, Z(37, 64), 0.5)[..., :5], ref[..., :5]) < NLM_TOL


@pytest.mark.parametrize("cfg", ["ref", "bench", "generic", "rt7", "rt5", "rt4", "rt2", "rt1", "rt10", "rt11", "rt12", "rt13", "rt16", "naive"])
def test_nlm_temporal_fused_equals_dispatch_sequence(ctx, cfg):
    """mid_nlm_temporal(k) == for each neighbour frame: mid_nlm_accum; then mid_normalize -- bit for
    bit -- and both match the oracle; windows clip at the sequence ends (5 frames, k=2)."""
    rng = np.random.default_rng(44)
    h, w = 30, 61
    base = (synth_hdr(rng, h, w) * 0.25).astype(np.float32)
    frames = [(np.roll(base, 2 * i, axis=1) * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32) for i in range(5)]
    fused = ctx.nlm_temporal(frames, k=2, **NLM_CFGS[cfg])
    ref = oracle.nlm_temporal(frames, k=2, **NLM_CFGS[cfg])
    for t in range(5):
        W = Z(h, w)
        for f in range(max(0, t - 2), min(4, t + 2) + 1):
            W = ctx.nlm_accum(frames[t], frames[f], W, 0.5, **NLM_CFGS[cfg])
        assert np.array_equal(fused[t], ctx.normalize(W))
        assert rel_err(fused[t], ref[t]) < NLM_TOL
    part = ctx.nlm_temporal(frames, k=2, first=1, count=3, **NLM_CFGS[cfg])
    assert all(np.array_equal(part[i], fused[1 + i]) for i in range(3)), "a sub-range sees the same halo frames"


# ---- a5 / a6 --------------------------------------------------------------------------------------
def test_normalize_bit_exact(ctx):
    rng = np.random.default_rng(50)
    W = (rng.random((67, 93, 8), dtype=np.float32) * 100).astype(np.float32)
    W[::7, ::5, 4] = 0.0                                     # magenta sentinel
    W[3, 3] = [np.inf, -1, 0, 1e-30, 1e-30, 7, 7, 7]
    assert np.array_equal(ctx.normalize(W), oracle.normalize(W))


def test_u8_paths_bit_exact(ctx):
    rng = np.random.default_rng(51)
    u8 = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 4096 + 3, dtype=np.uint8)])
    for fl in (0, 1):
        assert np.array_equal(ctx.unpack_u8(u8, fl), oracle.unpack_u8(u8, fl))
    f = np.concatenate([(rng.random(8191, dtype=np.float32) * 1.3 - 0.15),
                        np.float32([0, 1, -0.0, 255.999 / 255, 256.0 / 255, -1 / 255, -1.0001 / 255, np.nan, np.inf, -np.inf]),
                        np.nextafter(np.arange(1, 256, dtype=np.float32) / np.float32(255), np.float32(0))]).astype(np.float32)
    assert np.array_equal(ctx.pack_u8(f), oracle.pack_u8(f))


# ---- golden fixtures ------------------------------------------------------------------------------
def test_against_committed_golden_vectors(ctx):
    g = np.load(os.path.join(GOLDEN, "shader_restatement.npz"))
    hdr, ldr = g["hdr"], g["ldr"]
    h, w = hdr.shape[:2]
    assert rel_err(ctx.bilateral(hdr, 4, 2.0, 0.2, "texture"), g["bil_tex_r4"]) < BIL_TOL
    assert rel_err(ctx.bilateral(hdr, 4, 2.0, 0.2, "linear"), g["bil_lin_r4"]) < BIL_TOL
    assert rel_err(ctx.bilateral(ldr, 8, 2.0, 0.2, "texture"), g["bil_tex_r8_ldr"]) < BIL_TOL
    assert rel_err(ctx.bilateral_layers(hdr, [g["layer0"], g["layer1"]], 4), g["layers_out"]) < BIL_TOL
    assert rel_err(ctx.nlm_accum(g["nlm_in_t"], g["nlm_in_n"], Z(h, w), 0.5, (-7, 7), (-3, 3))[..., :5], g["nlm_ref_W"][..., :5]) < NLM_TOL
    assert rel_err(ctx.nlm_accum(g["nlm_in_t"], g["nlm_in_n"], Z(h, w), 0.5, (-10, 11), (-3, 4))[..., :5], g["nlm_bench_W"][..., :5]) < NLM_TOL
    assert np.array_equal(ctx.normalize(g["layers_W"]), g["layers_out"])


def test_gpu_bilateral_vs_reference_cpu_loop_golden(ctx):
    """The reference's CPU path and its GPU shaders are different filters only through the blue-channel
    typo, the alpha=1 override and the border; with blue made constant, the interior RGB of the
    GPU linear kernel must equal the reference CPU loop's golden output (config 1 plumbing)."""
    g = np.load(os.path.join(GOLDEN, "ref_cpu_bilateral_b.npz"))
    img, R = g["img"].copy(), int(g["radius"])
    img[..., 2] = 0.25
    ref = oracle.cpu_bilateral(img, R, 10.0, 0.2, True, 1)
    out = ctx.bilateral(img, R, 10.0, 0.2, "linear")
    assert rel_err(out[R:-R, R:-R, :3], ref[R:-R, R:-R, :3]) < BIL_TOL


# ---- error behaviour ------------------------------------------------------------------------------
def test_invalid_arguments_are_errors_not_crashes(ctx):
    img = np.zeros((8, 8, 4), np.float32)
    with pytest.raises(mid.MidError) as e:
        ctx.bilateral(img, 0)
    assert e.value.code == 1
    with pytest.raises(mid.MidError):
        ctx.bilateral(img, 25)
    with pytest.raises(mid.MidError):
        ctx.bilateral(img, 4, sigma_s=0.0)
    with pytest.raises(mid.MidError):
        ctx.nlm_accum(img, img, Z(8, 8), 0.5, search=(1, 3), patch=(-1, 2))      # range must contain 0
    with pytest.raises(mid.MidError):
        ctx.nlm_accum(img, img, Z(8, 8), -1.0)
    with pytest.raises(mid.MidError):
        ctx.nlm_temporal([img, img], k=1, first=1, count=2)                      # past the end
    import ctypes
    p = mid.BilateralParams(8, 8, 2.0, 0.2, 4, 1, 0)                             # linear layout + layers: illegal,
    d = ctx.upload(img)                                                          # src/main.cpp:1406-1428
    assert mid.lib.mid_bilateral_layers_accum(ctx.handle, ctypes.byref(p), d.ptr, d.ptr, d.ptr, None) == 1
    assert mid.lib.mid_bilateral(ctx.handle, ctypes.byref(p), d.ptr, d.ptr, None) == 1   # in-place


# ---- sizes beyond one kernarg frame table / beyond 1080p ------------------------------------------
def test_nlm_temporal_chunks_long_sequences(ctx):
    """More than 96 frames: mid_nlm_temporal splits the outputs into chunks that carry their own halo."""
    rng = np.random.default_rng(60)
    frames = [rng.random((6, 9, 4), dtype=np.float32) for _ in range(110)]
    got = ctx.nlm_temporal(frames, k=2, search=(-2, 3), patch=(-1, 2))
    ref = oracle.nlm_temporal(frames, k=2, search=(-2, 3), patch=(-1, 2))
    assert max(rel_err(a, b) for a, b in zip(got, ref)) < NLM_TOL
    got_b = ctx.nlm_temporal(frames, k=2, **NLM_CFGS["bench"])
    for t in (0, 1, 46, 47, 48, 93, 94, 95, 109):               # around the chunk seams (92 outputs per launch at k=2)
        W = Z(6, 9)
        for f in range(max(0, t - 2), min(109, t + 2) + 1):
            W = ctx.nlm_accum(frames[t], frames[f], W, 0.5, **NLM_CFGS["bench"])
        assert np.array_equal(got_b[t], ctx.normalize(W)), t


def test_4k_frame_windows(ctx):
    """3840x2160: index arithmetic beyond 1080p; oracle on windows around the far corner and a seam."""
    rng = np.random.default_rng(61)
    H4, W4 = 2160, 3840
    img = (rng.random((H4, W4, 4), dtype=np.float32) * 2).astype(np.float32)
    out_b = ctx.bilateral(img, 8, 2.0, 0.2, "texture")
    out_n = ctx.nlm_temporal([img], k=0, **NLM_CFGS["ref"])[0]
    for y0, x0 in ((H4 - 20, W4 - 20), (1000, 3700), (2100, 0)):
        for out, halo, fn in ((out_b, 8, lambda c: oracle.bilateral_texture(c, 8, 2.0, 0.2)),
                              (out_n, 10, lambda c: oracle.normalize(oracle.nlm_accum(c, c, Z(*c.shape[:2]), 0.5)))):
            ya, yb, xa, xb = y0 - halo, y0 + 20 + halo, x0 - halo, x0 + 20 + halo
            c = np.zeros((yb - ya, xb - xa, 4), np.float32)
            sy, sx = slice(max(ya, 0), min(yb, H4)), slice(max(xa, 0), min(xb, W4))
            c[sy.start - ya:sy.stop - ya, sx.start - xa:sx.stop - xa] = img[sy, sx]
            assert rel_err(out[y0:y0 + 20, x0:x0 + 20], fn(c)[halo:halo + 20, halo:halo + 20]) < NLM_TOL, (y0, x0)


# ---- non-finite texels (EXR renders do contain them) -----------------------------------------------
def test_nan_and_inf_texels_propagate_like_the_oracle(ctx):
    """A NaN texel poisons exactly the pixels whose windows see it; +inf texels follow IEEE (inf-inf, 0*inf = NaN).
    The set of non-finite outputs must match the oracle's and the finite ones stay within tolerance."""
    rng = np.random.default_rng(70)
    h, w = 60, 100
    img = (synth_hdr(rng, h, w) * 0.25).astype(np.float32)
    img[20, 30, 1] = np.nan
    img[40, 70, 0] = np.inf
    img[5, 90, 2] = -np.inf
    img[50, 10, 3] = np.nan                       # alpha only: not part of any distance
    for name, got, ref in (
        ("bilateral", ctx.bilateral(img, 8, 2.0, 0.2), oracle.bilateral_texture(img, 8, 2.0, 0.2)),
        ("nlm", ctx.nlm_temporal([img], k=0, **NLM_CFGS["bench"])[0], oracle.nlm_temporal([img], k=0, **NLM_CFGS["bench"])[0]),
    ):
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name
        assert np.array_equal(np.isinf(got), np.isinf(ref)), name
        fin = np.isfinite(ref)
        assert fin.mean() > 0.5 and rel_err(got[fin], ref[fin]) < NLM_TOL, name


# ---- degenerate and very large frames ---------------------------------------------------------------
def test_empty_frames_are_rejected(ctx):
    """A 0-pixel frame is an argument error (code 1), for every operator."""
    import ctypes
    d = ctx.alloc(64)
    for w_, h_ in ((0, 8), (8, 0), (0, 0), (-3, 4)):
        bp = mid.BilateralParams(w_, h_, 2.0, 0.2, 4, 0, 0)
        npar = mid.NlmParams(w_, h_, 0.5, -7, 7, -3, 3, 0)
        zp = mid.NormalizeParams(w_, h_)
        assert mid.lib.mid_bilateral(ctx.handle, ctypes.byref(bp), d.ptr, ctx.alloc(64).ptr, None) == 1
        assert mid.lib.mid_nlm_accum(ctx.handle, ctypes.byref(npar), d.ptr, d.ptr, d.ptr, None) == 1
        assert mid.lib.mid_normalize(ctx.handle, ctypes.byref(zp), d.ptr, d.ptr, None) == 1
    assert mid.lib.mid_pack_u8(ctx.handle, d.ptr, 0, d.ptr, None) == 0          # zero values: nothing to do, not an error


def test_frame_larger_than_2_gib(ctx):
    """16384 x 8704 RGBA32F = 2.28 GB per buffer: byte offsets pass 2^31, pixel indices stay in int range.
    Bilateral r=4 (tuned tile) checked on windows at the far corner; pack/unpack over the whole buffer."""
    H5, W5 = 8704, 16384
    rng = np.random.default_rng(80)
    base = rng.random((256, 512, 4), dtype=np.float32)
    img = np.tile(base, (H5 // 256, W5 // 512, 1))
    img[-64:, -64:] = rng.random((64, 64, 4), dtype=np.float32) * 2
    d_in, d_out = ctx.upload(img), ctx.alloc(img.nbytes)
    ctx.bilateral_dev(d_in.ptr, d_out.ptr, W5, H5, 4, 2.0, 0.2, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)
    out = ctx.download(d_out, (H5, W5, 4), np.float32)
    for y0, x0 in ((H5 - 24, W5 - 24), (H5 - 24, 0), (4000, W5 - 24), (0, 0)):
        ya, yb, xa, xb = y0 - 4, y0 + 28, x0 - 4, x0 + 28
        c = np.zeros((32, 32, 4), np.float32)
        sy, sx = slice(max(ya, 0), min(yb, H5)), slice(max(xa, 0), min(xb, W5))
        c[sy.start - ya:sy.stop - ya, sx.start - xa:sx.stop - xa] = img[sy, sx]
        assert rel_err(out[y0:y0 + 24, x0:x0 + 24], oracle.bilateral_texture(c, 4, 2.0, 0.2)[4:28, 4:28]) < BIL_TOL, (y0, x0)
    import ctypes
    d_u8 = ctx.alloc(H5 * W5 * 4)
    assert mid.lib.mid_pack_u8(ctx.handle, d_in.ptr, H5 * W5 * 4, d_u8.ptr, None) == 0
    u8 = ctx.download(d_u8, (H5, W5, 4), np.uint8)
    assert np.array_equal(u8[-300:], oracle.pack_u8(img[-300:])) and np.array_equal(u8[:300], oracle.pack_u8(img[:300]))


# ---- randomized differential sweep ------------------------------------------------------------------
def test_random_parameter_sweep_against_oracle(ctx):
    """40 random (size, format, operator, parameter) draws, every one compared with the oracle.
    (MID_SWEEP_CASES / MID_SWEEP_SEED lengthen or re-seed the sweep for a soak run.)"""
    rng = np.random.default_rng(int(os.environ.get("MID_SWEEP_SEED", "2025")))
    for case in range(int(os.environ.get("MID_SWEEP_CASES", "40"))):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 150))
        ldr = bool(rng.integers(0, 2))
        img = synth_ldr(rng, h, w) if ldr else (synth_hdr(rng, h, w) * float(rng.uniform(0.1, 1.0))).astype(np.float32)
        f32 = oracle.unpack_u8(img, 0) if ldr else img
        op = int(rng.integers(0, 4))
        if op == 0:
            R = int(rng.choice([1, 2, 4, 5, 8, 9, 10, 12]))
            ss, sc = float(rng.uniform(0.8, 6.0)), float(rng.uniform(0.05, 1.0))
            lay = ["texture", "linear"][int(rng.integers(0, 2))]
            ref = (oracle.bilateral_texture if lay == "texture" else oracle.bilateral_linear)(f32, R, ss, sc)
            err, tol = rel_err(ctx.bilateral(img, R, ss, sc, lay), ref), BIL_TOL
        elif op == 1:
            R = int(rng.choice([2, 4, 8]))
            layers = [synth_ldr(rng, h, w) for _ in range(int(rng.integers(1, 4)))]
            Wo = Z(h, w)
            for l in layers:
                Wo = oracle.bilateral_layers_accum(f32, l, Wo, R, 2.0, 0.2)
            err, tol = rel_err(ctx.bilateral_layers(img, layers, R), oracle.normalize(Wo)), BIL_TOL
        else:
            cfg = [NLM_CFGS["ref"], NLM_CFGS["bench"], dict(search=(-4, 5), patch=(-2, 3)), dict(search=(-3, 2), patch=(-1, 2)),
                   dict(search=(-5, 6), patch=(-3, 3))][int(rng.integers(0, 5))]
            hp = float(rng.uniform(0.2, 1.5))
            other = synth_ldr(rng, h, w) if ldr else (img * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32)
            o32 = oracle.unpack_u8(other, 0) if ldr else other
            if op == 2:
                W0 = rng.random((h, w, 8), dtype=np.float32)
                err = rel_err(ctx.nlm_accum(img, other, W0, hp, **cfg)[..., :5], oracle.nlm_accum(f32, o32, W0, hp, **cfg)[..., :5])
            else:
                got = ctx.nlm_temporal([img, other], k=1, hparam=hp, **cfg)
                ref = oracle.nlm_temporal([f32, o32], k=1, hparam=hp, **cfg)
                err = max(rel_err(a, b) for a, b in zip(got, ref))
            tol = NLM_TOL
        assert err < tol, (case, op, h, w, ldr, err)


def test_runtime_range_kernel_small_then_large_window(ctx):
    """The same run-time-range kernel first with a small LDS tile, then with a large one (its dynamic-LDS
    limit must not stay at the first call's size)."""
    rng = np.random.default_rng(90)
    t = rng.random((40, 70, 4), dtype=np.float32)
    nb = np.clip(t + 0.05 * rng.standard_normal((40, 70, 4)), 0, 1).astype(np.float32)
    for search in ((-2, 3), (-14, 15), (-3, 4), (-20, 21)):
        got = ctx.nlm_accum(t, nb, Z(40, 70), 0.5, search=search, patch=(-2, 3))
        # up to 1681 offsets: held against the float64 restatement (two fp32 evaluations may differ by twice the tolerance)
        assert rel_err(got[..., :5], _nlm_f64(t, nb, Z(40, 70), 0.5, search, (-2, 3))) < NLM_TOL, search


@pytest.mark.parametrize("search,patch", [((-12, 13), (-3, 4)), ((-15, 16), (-1, 2)), ((-11, 12), (-2, 3)), ((-17, 18), (-2, 3))])
def test_nlm_large_window_eight_wave_workgroups(ctx, search, patch):
    """Windows whose 4-wave tile no longer fits twice into a CU's LDS run on 8-wave workgroups (64-row tiles).  Strips
    are still 8 rows at multiples of 8, so a pixel's bits may depend neither on the workgroup shape nor on where the
    pixel sits in its tile: fused == accumulate + normalize, the result is invariant under a shift of the frame by 8
    rows and 5 columns (other wave, other tile, other lane), and it matches the oracle -- on a frame taller than one
    64-row tile, ragged in both directions."""
    rng = np.random.default_rng(search[1] * 7 + patch[1])
    h, w = 150, 75
    t = (synth_hdr(rng, h, w) * 0.25).astype(np.float32)
    fused = ctx.nlm_temporal([t], k=0, search=search, patch=patch)[0]
    W = ctx.nlm_accum(t, t, Z(h, w), 0.5, search=search, patch=patch)
    assert np.array_equal(fused, ctx.normalize(W))
    ys, xs = slice(60, 90), slice(20, 60)                      # the oracle on a window across the 64-row tile seam
    halo = -search[0] + max(-patch[0], patch[1])
    c = np.zeros((30 + 2 * halo, 40 + 2 * halo, 4), np.float32)
    ya, xa = ys.start - halo, xs.start - halo
    sy, sx = slice(max(ya, 0), min(ya + c.shape[0], h)), slice(max(xa, 0), min(xa + c.shape[1], w))
    c[sy.start - ya:sy.stop - ya, sx.start - xa:sx.stop - xa] = t[sy, sx]
    ref = oracle.normalize(oracle.nlm_accum(c, c, Z(*c.shape[:2]), 0.5, search=search, patch=patch))[halo:halo + 30, halo:halo + 40]
    assert rel_err(fused[ys, xs], ref) < NLM_TOL
    shifted = ctx.nlm_temporal([np.ascontiguousarray(t[8:, 5:])], k=0, search=search, patch=patch)[0]
    assert np.array_equal(shifted[halo:-halo, halo:-halo], fused[8 + halo:-halo, 5 + halo:-halo]), "translation by (8 rows, 5 columns)"
    # temporal, three frames
    frames = [t, np.roll(t, 3, axis=1), np.roll(t, -2, axis=0)]
    fz = ctx.nlm_temporal(frames, k=1, search=search, patch=patch)
    for i in range(3):
        Wi = Z(h, w)
        for f in range(max(0, i - 1), min(2, i + 1) + 1):
            Wi = ctx.nlm_accum(frames[i], frames[f], Wi, 0.5, search=search, patch=patch)
        assert np.array_equal(fz[i], ctx.normalize(Wi)), i


@pytest.mark.parametrize("patch", [(-5, 5), (-5, 6), (-6, 6), (-6, 7), (-7, 7), (-7, 8), (-8, 8)])
def test_nlm_large_patches_four_row_strips(ctx, patch):
    """Patches of 10x10 .. 16x16 run on strips of four rows: the oracle on a frame taller than one 16-row tile and ragged
    in both directions, fused == accumulate + normalize, and a pixel's bits do not depend on where it sits (shift by 4
    rows and 3 columns: other wave, other tile, other lane)."""
    rng = np.random.default_rng(patch[1] - patch[0])
    search, h, w = (-6, 7), 45, 70
    t = (synth_hdr(rng, h, w) * 0.25).astype(np.float32)
    nb = (t * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32)
    Wg = ctx.nlm_accum(t, nb, Z(h, w), 0.5, search=search, patch=patch)
    assert rel_err(Wg[..., :5], oracle.nlm_accum(t, nb, Z(h, w), 0.5, search=search, patch=patch)[..., :5]) < NLM_TOL
    fused = ctx.nlm_temporal([t], k=0, search=search, patch=patch)[0]
    assert np.array_equal(fused, ctx.normalize(ctx.nlm_accum(t, t, Z(h, w), 0.5, search=search, patch=patch)))
    halo = -search[0] + max(-patch[0], patch[1])
    shifted = ctx.nlm_temporal([np.ascontiguousarray(t[4:, 3:])], k=0, search=search, patch=patch)[0]
    assert np.array_equal(shifted[halo:-halo, halo:-halo], fused[4 + halo:-halo, 3 + halo:-halo])


def test_impulse_response_against_the_float64_restatement(ctx):
    """One bright texel off-centre in a flat frame (and an impulse in the neighbour frame only, for NLM): the response is
    the filter's kernel, so a transposed offset, a mirrored window or a wrong window edge shows as a displaced or
    truncated footprint.  Compared with the float64 NumPy restatement (tests/np_reference.py), which shares no code with
    the oracle or the kernels; lopsided ranges on purpose."""
    import np_reference as npr
    h, w = 41, 53
    img = np.full((h, w, 4), 0.25, np.float32)
    img[..., 3] = 1.0
    img[17, 31, :3] = (0.9, 0.35, 0.6)                       # (y, x) = (17, 31): not on a diagonal of the frame
    for R, ss, sc in ((4, 2.0, 0.6), (8, 3.0, 0.8), (7, 2.5, 0.7)):
        num, den = npr.bilateral_texture(img, R, ss, sc)
        ref = num / den[..., None]
        got = ctx.bilateral(img, R, ss, sc, "texture")
        assert rel_err(got, ref) < 1e-5, R
        # the footprint itself: in the interior the impulse is felt exactly inside its (2R+1)^2 window
        flat = got[R + 2, R + 2, 0]
        felt = np.abs(got[2 * R:h - 2 * R, 2 * R:w - 2 * R, 0] - flat) > 1e-6
        ys, xs = np.nonzero(felt)
        assert ys.min() + 2 * R >= 17 - R and ys.max() + 2 * R <= 17 + R and xs.min() + 2 * R >= 31 - R and xs.max() + 2 * R <= 31 + R
        assert rel_err(ctx.bilateral(img, R, ss, sc, "linear"), npr.bilateral_linear(img, R, ss, sc)) < 1e-5, R
    t = np.full((h, w, 4), 0.25, np.float32); t[..., 3] = 1.0
    nb = t.copy(); nb[17, 31, :3] = (0.5, 0.3, 0.4)
    for search, patch in (((-3, 6), (-2, 3)), ((-7, 7), (-3, 3)), ((-5, 2), (-1, 2)), ((-4, 5), (-5, 6))):
        num, den = npr.nlm_sums(t, nb, 0.5, search, patch)
        W = ctx.nlm_accum(t, nb, Z(h, w), 0.5, search=search, patch=patch)
        assert rel_err(W[..., :4], num) < NLM_TOL and rel_err(W[..., 4], den) < NLM_TOL, (search, patch)


@pytest.mark.parametrize("R", list(range(1, 25)))
def test_bilateral_every_radius(ctx, R):
    """radius is a run-time parameter of the ABI (1..24): tuned tiles, the run-time-radius tiled kernel and the
    per-pixel fallback between them must cover the whole range."""
    rng = np.random.default_rng(100 + R)
    img = synth_hdr(rng, 30, 70)
    for layout, f in (("texture", oracle.bilateral_texture), ("linear", oracle.bilateral_linear)):
        assert rel_err(ctx.bilateral(img, R, 1.0 + R / 4, 0.3, layout), f(img, R, 1.0 + R / 4, 0.3)) < BIL_TOL, (R, layout)


@pytest.mark.parametrize("search,patch", [((-5, 9), (-3, 4)), ((-12, 13), (-2, 3)), ((0, 1), (-3, 3)), ((-1, 2), (-4, 5)),
                                          ((-16, 17), (-1, 2)), ((-3, 1), (-1, 2)), ((-9, 10), (0, 1)), ((-2, 3), (-5, 6))])
def test_nlm_unusual_windows(ctx, search, patch):
    """Asymmetric and degenerate search ranges, patch sizes the run-time-range kernel knows (11x11: the four-row strips)."""
    rng = np.random.default_rng(abs(search[0]) * 31 + patch[1])
    t, nb = synth_hdr(rng, 29, 71) * 0.3, synth_hdr(rng, 29, 71) * 0.3
    got = ctx.nlm_accum(t, nb, Z(29, 71), 0.45, search, patch)
    # Held against float64 (see test_nlm_accum, "rt5").  The tolerance 2e-5 is SURVEY.md 8c's figure for the windows it
    # names (<= 21x21 = 441 offsets); an fp32 sum of N non-negative terms is only good to about N * 2^-24 of its value
    # (terms below half an ulp of the running sum are dropped one by one), in the shader's own order too: at 33x33 =
    # 1089 offsets oracle.c, which adds in nonlocal.comp's order, is itself 2.5e-5 from the exact sums on this very
    # input (the same 1089 fp32 weights added in float64 are 1.4e-7 away), the kernel 3.2e-5.  So beyond 441 offsets
    # the tolerance grows with the number of terms, for the kernel and for the reference order alike.
    n_off = (search[1] - search[0]) ** 2
    tol = NLM_TOL * max(1.0, n_off / 441.0)
    ref = _nlm_f64(t, nb, Z(29, 71), 0.45, search, patch)
    assert rel_err(got[..., :5], ref) < tol, (search, patch)
    assert rel_err(oracle.nlm_accum(t, nb, Z(29, 71), 0.45, search, patch)[..., :5], ref) < tol, (search, patch)
    assert not got[..., 5:].any()


# ---- batched plain bilateral ------------------------------------------------------------------------
@pytest.mark.parametrize("radius,layout", [(8, "linear"), (8, "texture"), (4, "texture"), (6, "linear"), (20, "texture")])
def test_bilateral_batch_is_the_single_frame_call_repeated(ctx, radius, layout):
    """mid_bilateral_batch (one launch, grid = tiles x frames) == mid_bilateral per frame, bit for bit: tuned radii,
    a run-time radius (6) and both addressing rules; one frame is checked against the oracle."""
    rng = np.random.default_rng(700 + radius)
    h, w = 53, 141
    frames = [synth_hdr(rng, h, w) for _ in range(5)]
    got = ctx.bilateral_batch(frames, radius, 2.0, 0.2, layout)
    for f, g in zip(frames, got):
        assert np.array_equal(g, ctx.bilateral(f, radius, 2.0, 0.2, layout))
    ref = (oracle.bilateral_linear if layout == "linear" else oracle.bilateral_texture)(frames[3], radius, 2.0, 0.2)
    assert rel_err(got[3], ref) < BIL_TOL
    ldr = [synth_ldr(rng, 40, 70) for _ in range(3)]
    got8 = ctx.bilateral_batch(ldr, radius, 2.0, 0.2, layout)
    assert all(np.array_equal(g, ctx.bilateral(f, radius, 2.0, 0.2, layout)) for f, g in zip(ldr, got8))


def test_bilateral_batch_argument_errors(ctx):
    import ctypes
    img = np.zeros((8, 8, 4), np.float32)
    d = ctx.upload(img)
    o = ctx.alloc(8 * 8 * 16)
    p = mid.BilateralParams(8, 8, 2.0, 0.2, 4, 0, 0)
    one = (ctypes.c_void_p * 1)(d.ptr)
    out = (ctypes.c_void_p * 1)(o.ptr)
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), one, out, 0, None) == 1       # n_frames < 1
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), one, one, 1, None) == 1       # in-place
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), None, out, 1, None) == 1
    nul = (ctypes.c_void_p * 1)(None)
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), nul, out, 1, None) == 1
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), one, out, 1, None) == 0
    ctx.sync()
    # ping-pong tables shifted by one slot: out[0] is in[1] -- every frame of a launch runs concurrently, so that is a race
    d2, o2 = ctx.upload(img), ctx.alloc(8 * 8 * 16)
    ins = (ctypes.c_void_p * 2)(d.ptr, d2.ptr)
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), ins, (ctypes.c_void_p * 2)(d2.ptr, o2.ptr), 2, None) == 1
    assert b"also an input" in mid.lib.mid_last_error()
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), ins, (ctypes.c_void_p * 2)(o.ptr, o.ptr), 2, None) == 1   # one buffer twice
    assert mid.lib.mid_bilateral_batch(ctx.handle, ctypes.byref(p), ins, (ctypes.c_void_p * 2)(o.ptr, o2.ptr), 2, None) == 0
    ctx.sync()


@pytest.mark.parametrize("search,patch", [((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))])
def test_nlm_known_answers_for_frames_that_vary_in_both_axes(ctx, search, patch):
    """The NLM kernels (fused temporal k = 1, and the single-frame launch with its HALF tail) on frames colour(x, y) = f(x) + g(y) against the
    known answers worked out by hand from the shader's text (tests/np_reference.py::nlm_additive_known_answer: 1-D sums + a cross term, no
    image loops): 160 x 240 = 5 x 5 tiles, every pixel whose window and patch stay inside the image."""
    from conftest import additive_frames
    from np_reference import nlm_additive_known_answer
    rng = np.random.default_rng(19)
    h, w = 160, 240
    frs = additive_frames(rng, h, w, 3)
    m = max(-search[0], search[1] - 1) + max(-patch[0], patch[1] - 1)
    want = nlm_additive_known_answer(frs[1][0], frs[1][1], 0.5, search, patch, neighbours=[(a, b) for a, b, _ in frs])
    got = ctx.nlm_temporal([x[2] for x in frs], k=1, first=1, count=1, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got[m:-m, m:-m], want[m:-m, m:-m]) < 2e-5
    want1 = nlm_additive_known_answer(frs[0][0], frs[0][1], 0.5, search, patch)
    got1 = ctx.nlm_temporal([frs[0][2]], k=0, hparam=0.5, search=search, patch=patch)[0]
    assert rel_err(got1[m:-m, m:-m], want1[m:-m, m:-m]) < 2e-5
    assert np.abs(want1[m:-m, m:-m, :3] - frs[0][2][m:-m, m:-m, :3]).max() > 0.03


def test_nlm_temporal_and_layers_refuse_aliased_outputs(ctx):
    """Every output frame of a launch is computed concurrently from the frames around it: an output that is a frame of the sequence (in place,
    ping-pong tables shifted by a slot) or a buffer given twice is refused before anything is launched, like mid_bilateral_batch."""
    import ctypes
    img = np.zeros((8, 8, 4), np.float32)
    fr = [ctx.upload(img) for _ in range(3)]
    o = [ctx.alloc(8 * 8 * 16) for _ in range(3)]
    p = mid.NlmParams(8, 8, 0.5, -7, 7, -3, 3, mid.FMT_RGBA32F)
    frames = (ctypes.c_void_p * 3)(*[f.ptr for f in fr])

    def call(outs, first=0, count=3, k=1):
        return mid.lib.mid_nlm_temporal(ctx.handle, ctypes.byref(p), frames, 3, k, first, count, (ctypes.c_void_p * len(outs))(*outs), None)
    assert call([fr[0].ptr, o[1].ptr, o[2].ptr]) == 1 and b"also a frame of the sequence" in mid.lib.mid_last_error()      # in place
    assert call([fr[1].ptr], first=0, count=1) == 1                                     # out[0] is the NEXT frame: a shifted ping-pong table
    assert call([o[0].ptr, o[0].ptr, o[2].ptr]) == 1 and b"appears twice" in mid.lib.mid_last_error()
    assert call([o[0].ptr, o[1].ptr, o[2].ptr]) == 0
    ctx.sync()
    bp = mid.BilateralParams(8, 8, 2.0, 0.2, 4, 0, 0)
    lay = ctx.upload(np.zeros((8, 8, 4), np.uint8))
    tbl = (ctypes.c_void_p * 1)(lay.ptr)
    assert mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), fr[0].ptr, tbl, 1, fr[0].ptr, None) == 1 and b"in-place" in mid.lib.mid_last_error()
    assert mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), fr[0].ptr, tbl, 1, o[0].ptr, None) == 0
    ctx.sync()
    # the streaming passes change the pixel stride: never in place
    zp = mid.NormalizeParams(8, 8)
    Wb = ctx.alloc(8 * 8 * 32)
    assert mid.lib.mid_normalize(ctx.handle, ctypes.byref(zp), Wb.ptr, Wb.ptr, None) == 1 and b"WeightInfo" in mid.lib.mid_last_error()
    assert mid.lib.mid_pack_u8(ctx.handle, o[0].ptr, 8 * 8 * 4, o[0].ptr, None) == 1 and b"in place" in mid.lib.mid_last_error()
    assert mid.lib.mid_unpack_u8(ctx.handle, o[0].ptr, 8 * 8 * 4, 0, o[0].ptr, None) == 1


def test_bilateral_batch_chunks_beyond_the_frame_table(ctx):
    """More frames than the by-value frame table holds (96): several launches, same bits."""
    rng = np.random.default_rng(811)
    frames = [synth_hdr(rng, 20, 70) for _ in range(101)]
    got = ctx.bilateral_batch(frames, 4, 2.0, 0.2, "texture")
    for i in (0, 1, 95, 96, 97, 100):
        assert np.array_equal(got[i], ctx.bilateral(frames[i], 4, 2.0, 0.2, "texture")), i


# ---- opaque-tile forms (round 6) -------------------------------------------------------------------
def test_opaque_and_general_tile_forms_agree(ctx):
    """Where every texel of a tile has alpha == 1.0f the kernels run a shorter loop (csrc/bilateral.hip: no alpha FMA;
    csrc/nlm_strip.hpp: no weight add, tuned windows).  One texel with alpha = 0.25 in the middle of the frame puts the tiles
    that see it on the general loop and leaves the others on the short one; alpha enters no weight, so
      * bilateral (plain and layer-guided): the rgb output of the two frames is the same BITS everywhere -- the forms are
        bit-identical -- and alpha differs only inside the texel's window;
      * NLM (bench and reference windows, single frame and temporal): rgb agrees to the last bit or two (the opaque form adds
        the 0.001 of nonlocal.comp:32 after the weights instead of before them);
      * both frames match the oracle, alpha included."""
    rng = np.random.default_rng(77)
    h, w = 150, 300                                   # 3 x 10 bilateral tiles of 64 x 16, 6 x 5 NLM tiles of 58 x 32: interior tiles exist
    a = (synth_hdr(rng, h, w) * 0.25).astype(np.float32)
    a[..., 3] = 1.0                                   # (the scale above applies to alpha too)
    b = a.copy()
    b[75, 150, 3] = 0.25
    for R in (4, 8):
        for layout in ("texture", "linear"):
            ga, gb = ctx.bilateral(a, R, 2.0, 0.2, layout), ctx.bilateral(b, R, 2.0, 0.2, layout)
            assert np.array_equal(ga[..., :3], gb[..., :3]), (R, layout)
            diff = np.argwhere(ga[..., 3] != gb[..., 3])
            assert len(diff) and np.all(np.abs(diff - [75, 150]) <= R), (R, layout)
            ref = oracle.bilateral_texture(b, R, 2.0, 0.2) if layout == "texture" else oracle.bilateral_linear(b, R, 2.0, 0.2)
            assert rel_err(gb, ref) < BIL_TOL and np.all(ga[R:-R, R:-R, 3] == 1.0)
    layers = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(3)]
    la, lb = ctx.bilateral_layers(a, layers, 8, 2.0, 0.2), ctx.bilateral_layers(b, layers, 8, 2.0, 0.2)
    assert np.array_equal(la[..., :3], lb[..., :3]) and not np.array_equal(la[..., 3], lb[..., 3])
    Wl = Z(h, w)
    for lay in layers:
        Wl = oracle.bilateral_layers_accum(b, lay, Wl, 8, 2.0, 0.2)
    assert rel_err(lb, oracle.normalize(Wl)) < BIL_TOL
    for cfg in ("bench", "ref"):
        na, nb_ = ctx.nlm_temporal([a], k=0, **NLM_CFGS[cfg])[0], ctx.nlm_temporal([b], k=0, **NLM_CFGS[cfg])[0]
        assert rel_err(na[..., :3], nb_[..., :3]) < 5e-7, cfg                       # last-bit agreement of the two forms
        assert rel_err(nb_, oracle.nlm_temporal([b], k=0, **NLM_CFGS[cfg])[0]) < NLM_TOL
        assert rel_err(na, oracle.nlm_temporal([a], k=0, **NLM_CFGS[cfg])[0]) < NLM_TOL
        # temporal: the form is chosen per neighbour frame (frame 1 of the window is the one with the odd texel)
        seq_a, seq_b = [a, np.roll(a, 2, axis=1), a], [a, np.roll(b, 2, axis=1), a]
        ta, tb = ctx.nlm_temporal(seq_a, k=1, **NLM_CFGS[cfg]), ctx.nlm_temporal(seq_b, k=1, **NLM_CFGS[cfg])
        rb = oracle.nlm_temporal(seq_b, k=1, **NLM_CFGS[cfg])
        for t in range(3):
            assert rel_err(ta[t][..., :3], tb[t][..., :3]) < 5e-7 and rel_err(tb[t], rb[t]) < NLM_TOL, (cfg, t)
            W = Z(h, w)                                                             # fused == dispatch sequence, bit for bit, on mixed tiles too
            for f in range(max(0, t - 1), min(2, t + 1) + 1):
                W = ctx.nlm_accum(seq_b[t], seq_b[f], W, 0.5, **NLM_CFGS[cfg])
            assert np.array_equal(tb[t], ctx.normalize(W)), (cfg, t)
