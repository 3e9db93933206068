"""Test-only float64 evaluation of non-local means with torch -- a THIRD statement of nonlocal.comp:28-63 + normalize.comp:29-44.

It shares no code with oracle/oracle.c (fp32, the shader's loops: patch taps innermost), with tests/np_reference.py
(float64 NumPy, the same brute-force tap loop) or with the kernels (fp32 strips, block sums, DPP).  The form is the
textbook one: for every search offset s a squared-difference IMAGE D_s(q) = |T(q) - Nb(q+s)|^2_rgb over the zero-padded
frames, its PW x PW box sum by PW shifted adds per axis, exp, weighted sum of Nb(p+s); 0.001 is added to the norm once
per neighbour frame (nonlocal.comp:32), ranges are half-open (nonlocal.comp:36-44), out-of-image texels are vec4(0)
for both images.  Runs on the GPU box's device when torch sees one (a whole 1080p frame takes seconds),
on the CPU otherwise (small frames: tests/test_oracle.py holds it against oracle.c there).
"""
import numpy as np
import torch


def device():
    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


def _padded(img, P, dev):
    x = torch.as_tensor(np.ascontiguousarray(img, dtype=np.float32), device=dev).to(torch.float64)
    h, w, _ = x.shape
    out = torch.zeros((h + 2 * P, w + 2 * P, 4), dtype=torch.float64, device=dev)
    out[P:P + h, P:P + w] = x
    return out


def nlm_sums(target, neighbours, hparam, search, patch, dev=None):
    """(sum over neighbour frames and offsets of wt * Nb(p+s)  [h,w,4],  sum of (0.001 + sum wt)  [h,w]) in float64."""
    dev = device() if dev is None else dev
    h, w = target.shape[:2]
    slo, shi = search
    plo, phi = patch
    PW = phi - plo
    P = max(-plo, phi - 1) + max(-slo, shi - 1)
    tp = _padded(target, P, dev)
    num = torch.zeros((h, w, 4), dtype=torch.float64, device=dev)
    den = torch.zeros((h, w), dtype=torch.float64, device=dev)
    inv_h2 = 1.0 / (float(hparam) * float(hparam))
    # the rows/columns q of the difference image that some pixel's patch touches: [plo, h + phi - 1) x [plo, w + phi - 1)
    ya, yb, xa, xb = P + plo, P + h + phi - 1, P + plo, P + w + phi - 1
    a = tp[ya:yb, xa:xb, :3]
    for nb in neighbours:
        npd = _padded(nb, P, dev)
        den += 0.001
        for sy in range(slo, shi):
            for sx in range(slo, shi):
                b = npd[ya + sy:yb + sy, xa + sx:xb + sx, :3]
                D = ((a - b) ** 2).sum(-1)
                V = D[0:h]
                for j in range(1, PW):
                    V = V + D[j:j + h]
                B = V[:, 0:w]
                for i in range(1, PW):
                    B = B + V[:, i:i + w]
                wt = torch.exp(-B * inv_h2)
                num += npd[P + sy:P + sy + h, P + sx:P + sx + w] * wt[..., None]
                den += wt
    return num, den


def nlm_temporal_output(frames, t, k, hparam, search, patch, dev=None):
    """Output frame t of the multi-frame mode: neighbours max(0,t-k)..min(n-1,t+k), normalize.comp's division with its
    magenta sentinel for a zero norm; float64 NumPy array [h,w,4]."""
    lo, hi = max(0, t - k), min(len(frames) - 1, t + k)
    num, den = nlm_sums(frames[t], frames[lo:hi + 1], hparam, search, patch, dev)
    out = num / den[..., None]
    zero = den == 0
    if bool(zero.any()):
        out[zero] = torch.tensor([1.0, 0.0, 1.0, 1.0], dtype=torch.float64, device=out.device)
    return out.cpu().numpy()


def bilateral_sums(img, guide, R, sigma_s, sigma_c, linear=False, dev=None):
    """bialteral.comp:29-73 (linear=False: 2-D fetch, out-of-image texel = vec4(0)) / bialteral_linear.comp:29-72 (linear=True:
    flat index p + dx + dy*w, outside [0, N) = vec4(0), columns wrap into the adjacent row) / bialteral_layers.comp:27-62 (guide
    != img: range distance from the guide's rgb, colour from the image) in float64: (sum_q w c(q) [h,w,4], sum_q w [h,w]).
    `guide` is a float array [h,w,>=3] (an RGBA8 layer is passed as its UNORM decode)."""
    dev = device() if dev is None else dev
    h, w = img.shape[:2]
    n = h * w
    P = R * w + R                                        # flat padding covers every (dx, dy) of the window in both addressings
    def flat(a, c):
        x = torch.as_tensor(np.ascontiguousarray(a[..., :c], dtype=np.float32), device=dev).to(torch.float64).reshape(n, c)
        out = torch.zeros((n + 2 * P, c), dtype=torch.float64, device=dev)
        out[P:P + n] = x
        return out
    ip, gp = flat(img, 4), flat(guide, 3)
    g0 = gp[P:P + n]
    xs = torch.arange(n, device=dev) % w
    num = torch.zeros((n, 4), dtype=torch.float64, device=dev)
    den = torch.zeros((n,), dtype=torch.float64, device=dev)
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            o = P + dx + dy * w
            c, q = ip[o:o + n], gp[o:o + n]
            if not linear:                               # 2-D addressing: a column outside the row is out of the image, not the next row
                ok = ((xs + dx >= 0) & (xs + dx < w)).to(torch.float64)[:, None]
                c, q = c * ok, q * ok
            d2 = ((g0 - q) ** 2).sum(-1)
            wt = torch.exp(-0.5 * (dx * dx + dy * dy) / sigma_s ** 2 - 0.5 * d2 / sigma_c ** 2)
            num += c * wt[:, None]
            den += wt
    return num.reshape(h, w, 4), den.reshape(h, w)
