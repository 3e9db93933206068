"""GPU suite, maximum sizes: ONE frame whose buffers pass 4 GiB.

16400 x 16400 RGBA32F = 4,303,360,000 B per colour buffer (> 2^32), its WeightInfo buffer 8.6 GB: every kernel's byte
offsets leave 32 bits (row 16,368 starts beyond 4 GiB, row 8,184 beyond 2 GiB) while pixel indices stay inside `int`.
The frame is built on the device (no 4 GB host array); windows of the results are copied back and compared with the
oracle run on crops around them, as tests/test_gpu_fullsize.py does at 1080p.  Covered: the tuned bilateral tile in
both addressings (the linear one wraps rows, bialteral_linear.comp:58), the fused NLM strip kernel at the benchmark
window, mid_nlm_accum into the 8.6 GB weight buffer + mid_normalize, and pack/unpack over the whole buffer.
"""
import ctypes

import numpy as np
import pytest
import torch

import image_denoising_filter_amd as mid
import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu
H, W = 16400, 16400
SIZE = 20


def _crop(img_dev, y0, x0, halo):
    """Window + halo as a host array, zero beyond the frame (the texture policy)."""
    ya, yb, xa, xb = y0 - halo, y0 + SIZE + halo, x0 - halo, x0 + SIZE + halo
    out = np.zeros((yb - ya, xb - xa, 4), np.float32)
    sy, sx = slice(max(ya, 0), min(yb, H)), slice(max(xa, 0), min(xb, W))
    out[sy.start - ya:sy.stop - ya, sx.start - xa:sx.stop - xa] = img_dev[sy, sx].cpu().numpy()
    return out


# the far corner, the last rows at the left edge, the rows where byte offsets cross 2 GiB and 4 GiB, the origin
WINDOWS = ((H - SIZE, W - SIZE), (H - SIZE, 0), (8184 - 10, 5000), (16368 - 10, 8000), (16368 - 10, W - SIZE), (0, 0))


@pytest.fixture(scope="module")
def ts():
    """ONE explicit stream for torch's fills and the library's kernels alike.  (A NULL stream argument means the context's own
    compute stream, include/mi_denoise.h:15 -- passing torch's default-stream handle, which IS 0, would let a kernel start
    while the tensor it reads is still being written on torch's stream.)"""
    return torch.cuda.Stream(device=torch.device("cuda", 0))


@pytest.fixture(scope="module")
def huge(ts):
    assert H * W * 16 > 2 ** 32 and H * W < 2 ** 31
    dev = torch.device("cuda", 0)
    with torch.cuda.stream(ts):
        g = torch.Generator(device=dev).manual_seed(90)
        img = torch.rand((H, W, 4), device=dev, generator=g, dtype=torch.float32)
        img[..., :3] *= 0.8
    ts.synchronize()
    yield img
    del img
    torch.cuda.empty_cache()


def test_bilateral_both_addressings_beyond_4_gib(ctx, huge, ts):
    assert ts.cuda_stream != 0
    out = torch.empty_like(huge)
    for layout, orc in ((mid.LAYOUT_TEXTURE, oracle.bilateral_texture), (mid.LAYOUT_LINEAR, oracle.bilateral_linear)):
        ctx.bilateral_dev(huge.data_ptr(), out.data_ptr(), W, H, 8, 2.0, 0.2, layout, mid.FMT_RGBA32F, ts.cuda_stream)
        torch.cuda.synchronize()
        for y0, x0 in WINDOWS:
            if layout == mid.LAYOUT_LINEAR and (x0 < 8 or x0 + SIZE > W - 8):
                continue        # (row ends of the flat addressing see the neighbouring rows: covered at 1080p with whole rows)
            ref = orc(_crop(huge, y0, x0, 8), 8, 2.0, 0.2)[8:8 + SIZE, 8:8 + SIZE]
            assert rel_err(out[y0:y0 + SIZE, x0:x0 + SIZE].cpu().numpy(), ref) < 1e-5, (layout, y0, x0)
    # flat addressing at the very end of the buffer: the last rows, whole width, against the oracle on those rows
    rows = huge[H - 40:].cpu().numpy()
    ref = oracle.bilateral_linear(rows, 8, 2.0, 0.2)
    got = out[H - 40:].cpu().numpy()
    assert rel_err(got[10:, :64], ref[10:, :64]) < 1e-5 and rel_err(got[10:, -64:], ref[10:, -64:]) < 1e-5
    del out


def test_nlm_fused_and_accumulate_normalize_beyond_4_gib(ctx, huge, ts):
    search, patch, halo = (-10, 11), (-3, 4), 14
    s = ts.cuda_stream
    out = torch.empty_like(huge)
    ctx.nlm_temporal_dev([huge.data_ptr()], [out.data_ptr()], W, H, 0.5, search, patch, 0, 0, 1, mid.FMT_RGBA32F, s)
    # the unfused pair the reference dispatches: nonlocal.comp into the 32-byte-stride weight buffer, then normalize.comp
    with torch.cuda.stream(ts):
        Wb = torch.zeros((H, W, 8), device=huge.device, dtype=torch.float32)      # (the fill is ordered before the accumulate: same stream)
    assert Wb.numel() * 4 > 2 ** 33
    out2 = torch.empty_like(huge)
    p = mid.NlmParams(W, H, 0.5, search[0], search[1], patch[0], patch[1], mid.FMT_RGBA32F)
    assert mid.lib.mid_nlm_accum(ctx.handle, ctypes.byref(p), huge.data_ptr(), huge.data_ptr(), Wb.data_ptr(), s) == 0, mid.lib.mid_last_error()
    q = mid.NormalizeParams(W, H)
    assert mid.lib.mid_normalize(ctx.handle, ctypes.byref(q), Wb.data_ptr(), out2.data_ptr(), s) == 0, mid.lib.mid_last_error()
    torch.cuda.synchronize()
    for y0, x0 in WINDOWS:
        c = _crop(huge, y0, x0, halo)
        Wo = oracle.nlm_accum(c, c, np.zeros(c.shape[:2] + (8,), np.float32), 0.5, search=search, patch=patch)
        ref = oracle.normalize(Wo)[halo:halo + SIZE, halo:halo + SIZE]
        a = out[y0:y0 + SIZE, x0:x0 + SIZE].cpu().numpy()
        b = out2[y0:y0 + SIZE, x0:x0 + SIZE].cpu().numpy()
        assert rel_err(a, ref) < 2e-5, (y0, x0)
        assert np.array_equal(a, b), "fused == accumulate + normalize, bit for bit"
        wgot = Wb[y0:y0 + SIZE, x0:x0 + SIZE].cpu().numpy()
        assert rel_err(wgot[..., :5], Wo[halo:halo + SIZE, halo:halo + SIZE, :5]) < 2e-5, (y0, x0)
    # every pixel was written: the weight sums carry the 0.001 bias plus the zero-offset weight 1
    assert float(Wb[..., 4].min()) >= 1.0
    del Wb, out, out2


def test_pack_unpack_beyond_4_gib(ctx, huge, ts):
    s = ts.cuda_stream
    n = H * W * 4
    u8 = torch.empty((H, W, 4), device=huge.device, dtype=torch.uint8)
    assert mid.lib.mid_pack_u8(ctx.handle, huge.data_ptr(), n, u8.data_ptr(), s) == 0
    back = torch.empty_like(huge)
    assert mid.lib.mid_unpack_u8(ctx.handle, u8.data_ptr(), n, 0, back.data_ptr(), s) == 0
    torch.cuda.synchronize()
    for rows in (slice(0, 64), slice(8184 - 32, 8184 + 32), slice(H - 64, H)):
        h = huge[rows].cpu().numpy()
        assert np.array_equal(u8[rows].cpu().numpy(), oracle.pack_u8(h))
        assert np.array_equal(back[rows].cpu().numpy(), oracle.unpack_u8(oracle.pack_u8(h)))
    # whole buffer, on the device: pack(unpack(pack(x))) == pack(x)  (codes are fixed points of decode -> encode? not for
    # every code under truncation -- so the property checked is the weaker, exact one: the round trip never moves UP and
    # loses at most one code)
    u8b = torch.empty_like(u8)
    assert mid.lib.mid_pack_u8(ctx.handle, back.data_ptr(), n, u8b.data_ptr(), s) == 0
    torch.cuda.synchronize()
    d = u8.to(torch.int16) - u8b.to(torch.int16)
    assert int(d.min()) >= 0 and int(d.max()) <= 1
    del u8, u8b, back, d
