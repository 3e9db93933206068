import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """The product context on cuda:0.  No skip: on a GPU box a missing library/device is a failure."""
    import image_denoising_filter_amd as mid
    c = mid.Context(0)
    yield c
    c.close()


def rel_err(a, b):
    """max |a-b| / max(1,|b|): the tolerance form SURVEY.md 8c states."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def additive_frames(rng, h, w, n):
    """n frames colour(x, y) = f(x) + g(y) with f, g random walks on a 1/4096 grid, so that the sum is exact in fp32 and the float32 frame
    the filters see IS the additive frame the known answer is worked out for."""
    out = []
    for _ in range(n):
        f = np.round((np.cumsum(rng.normal(0, 0.02, (w, 3)), 0) + rng.uniform(0.2, 0.5, 3)) * 4096) / 4096
        g = np.round((np.cumsum(rng.normal(0, 0.02, (h, 3)), 0) + rng.uniform(0.1, 0.4, 3)) * 4096) / 4096
        img = np.concatenate([f[None, :, :] + g[:, None, :], np.ones((h, w, 1))], 2).astype(np.float32)
        assert np.array_equal(img[..., :3].astype(np.float64), f[None] + g[:, None])
        out.append((f, g, img))
    return out


def synth_ldr(rng, h, w):
    """8-bit test image: gradient + hard-edged discs + noise (scaled down)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([xx / max(w - 1, 1), yy / max(h - 1, 1), 0.5 + 0.5 * np.sin(xx * 0.3) * np.cos(yy * 0.2)], -1)
    for cx, cy, r, col in ((0.3, 0.4, 0.18, (0.9, 0.2, 0.1)), (0.7, 0.6, 0.22, (0.1, 0.8, 0.3))):
        m = (xx - cx * w) ** 2 + (yy - cy * h) ** 2 < (r * min(w, h)) ** 2
        img[m] = col
    img = img + rng.normal(0, 10 / 255, img.shape)
    rgba = np.concatenate([np.clip(img, 0, 1), np.ones((h, w, 1))], -1)
    return (rgba * 255).astype(np.uint8)


def synth_hdr(rng, h, w, scale=4.0):
    """HDR-range float image: smooth radiance + highlights + multiplicative Monte-Carlo-like noise."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([0.6 + 0.4 * np.sin(xx * 0.11 + 1.0), 0.5 + 0.5 * np.cos(yy * 0.07), 0.4 + 0.3 * np.sin((xx + yy) * 0.05)], -1)
    hl = np.exp(-(((xx - 0.6 * w) ** 2 + (yy - 0.3 * h) ** 2) / (0.02 * w * h + 1)))[..., None] * scale
    img = (base + hl) * rng.gamma(4.0, 0.25, (h, w, 1))
    a = np.ones((h, w, 1), np.float32)
    return np.concatenate([img, a], -1).astype(np.float32)
