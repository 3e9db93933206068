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
