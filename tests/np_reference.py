"""Independent float64 NumPy restatement of the shaders' formulas (not of their loop order).

Written from the maths in SURVEY.md 8a, vectorised over the image with shifted views, and used
to check oracle/oracle.c (fp32, reference loop order) from a second direction:
oracle.c vs this file must agree to fp32 rounding."""
import numpy as np


def _pad(img, p):
    h, w, c = img.shape
    out = np.zeros((h + 2 * p, w + 2 * p, c), np.float64)
    out[p:p + h, p:p + w] = img
    return out


def bilateral_texture(img, R, ss, sc, guide=None):
    """sum_q w c(q) and sum_q w, q over (2R+1)^2, zero texels outside the image."""
    img = img.astype(np.float64)
    g = img if guide is None else guide.astype(np.float64)
    h, w, _ = img.shape
    ip, gp = _pad(img, R), _pad(g, R)
    num = np.zeros((h, w, 4))
    den = np.zeros((h, w))
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            c = ip[R + dy:R + dy + h, R + dx:R + dx + w]
            q = gp[R + dy:R + dy + h, R + dx:R + dx + w]
            d2 = ((g[..., :3] - q[..., :3]) ** 2).sum(-1)
            wt = np.exp(-0.5 * (dx * dx + dy * dy) / ss ** 2) * np.exp(-0.5 * d2 / sc ** 2)
            num += c * wt[..., None]
            den += wt
    return num, den


def bilateral_linear(img, R, ss, sc):
    """Flat-index variant: texel (x+dx, y+dy) is flat[y*w + x + dx + dy*w], zero outside [0,N)."""
    img = img.astype(np.float64)
    h, w, _ = img.shape
    n = h * w
    flat = img.reshape(n, 4)
    idx = np.arange(n)
    num = np.zeros((n, 4))
    den = np.zeros(n)
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            j = idx + dx + dy * w
            ok = (j >= 0) & (j < n)
            q = np.where(ok[:, None], flat[np.clip(j, 0, n - 1)], 0.0)
            d2 = ((flat[:, :3] - q[:, :3]) ** 2).sum(-1)
            wt = np.exp(-0.5 * (dx * dx + dy * dy) / ss ** 2) * np.exp(-0.5 * d2 / sc ** 2)
            num += q * wt[:, None]
            den += wt
    return (num / den[:, None]).reshape(h, w, 4)


def nlm_sums(target, nb, hparam, search, patch):
    """(sum_s w Nb(p+s), 0.001 + sum_s w) with w = exp(-SSD_patch / h^2); ranges half-open."""
    t, n_ = target.astype(np.float64), nb.astype(np.float64)
    h, w, _ = t.shape
    P = max(-patch[0], patch[1]) + max(-search[0], search[1]) + 1
    tp, npad = _pad(t, P), _pad(n_, P)
    num = np.zeros((h, w, 4))
    den = np.full((h, w), 0.001)
    for sy in range(search[0], search[1]):
        for sx in range(search[0], search[1]):
            d = np.zeros((h, w))
            for j in range(patch[0], patch[1]):
                for i in range(patch[0], patch[1]):
                    a = tp[P + j:P + j + h, P + i:P + i + w, :3]
                    b = npad[P + sy + j:P + sy + j + h, P + sx + i:P + sx + i + w, :3]
                    d += ((a - b) ** 2).sum(-1)
            wt = np.exp(-d / hparam ** 2)
            num += npad[P + sy:P + sy + h, P + sx:P + sx + w] * wt[..., None]
            den += wt
    return num, den
