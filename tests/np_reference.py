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


def nlm_step_edge_known_answer(w, xe, A, B, hparam, search, patch, neighbours=None):
    """Closed form of shaders/nonlocal.comp for a VERTICAL STEP EDGE, derived by hand from the shader's text (no image loops, no
    oracle, no float64 checker): every pixel left of column xe has colour A, every other pixel colour B (alpha 1).  For a target in
    column x and the candidate at search offset (sx, sy) -- sx, sy in [search) (nonlocal.comp:36-38) -- the patch distance is a PLAIN SUM
    over (i, j) in [patch)^2 (:42-52) of |T(p+(i,j)) - N(c+(i,j))|^2 over rgb; along the edge nothing depends on the row, so
        d(sx) = PW * sum_{i in [patch)} |T(x+i) - N(x+sx+i)|^2_rgb,
    the weight is exp(-d / h^2) (:55), the same for all SW values of sy, and with the 0.001 bias of the norm (:32) -- once PER NEIGHBOUR
    FRAME, since every dispatch adds its own sums to WeightInfo (:61-62) -- and the division of normalize.comp:42
        out(x) = sum_f sum_sx SW w_f(sx) N_f(x+sx) / sum_f (0.001 + sum_sx SW w_f(sx)).
    neighbours: list of (A_f, B_f) step-edge frames with the same edge column (default: the target itself, the single-frame filter).
    Returns the (w, 4) float64 row of interior rows (valid for columns at least max(|search|) + max(|patch|) away from the left / right border)."""
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    nbs = [(A, B)] if neighbours is None else [(np.asarray(a, np.float64), np.asarray(b, np.float64)) for a, b in neighbours]
    slo, shi = search
    plo, phi = patch
    PW, SW = phi - plo, shi - slo
    col = lambda x, a, b: a if x < xe else b
    out = np.zeros((w, 4), np.float64)
    for x in range(w):
        acc, norm = np.zeros(4), 0.0
        for a, b in nbs:
            norm += 0.001
            for sx in range(slo, shi):
                d = PW * sum(float(((col(x + i, A, B)[:3] - col(x + sx + i, a, b)[:3]) ** 2).sum()) for i in range(plo, phi))
                wt = SW * np.exp(-d / (hparam * hparam))
                acc += wt * col(x + sx, a, b)
                norm += wt
        out[x] = acc / norm
    return out


def bilateral_layers_columns_known_answer(img_cols, layer_cols_u8, radius, sigma_s, sigma_c):
    """Closed form of shaders/bialteral_layers.comp (+ normalize.comp) for frames whose colours depend on the COLUMN only, derived by hand
    from the shader's text.  With curCoord = (x + i, y + j) (:44) the range weight -- taken on the LAYER (:45-49) -- depends on i alone and
    the spatial weight exp(-.5 (i^2 + j^2) / sigma_s^2) (:41-42) factors into es(i) es(j); the sum over j of es(j) multiplies the weighted
    colour sum (:53, colours from the INPUT image) and the weight sum (:54) alike, is the same for every layer, and cancels in
    normalize.comp:42 after all layers have been accumulated (:57-58, one dispatch per layer, src/main.cpp:1610-1623):
        out(x) = sum_L sum_i es(i) wr_L(x, i) I(x + i) / sum_L sum_i es(i) wr_L(x, i),    wr_L = exp(-.5 |L(x) - L(x+i)|^2_rgb / sigma_c^2),
    layers UNORM-decoded (the fp32 texel c / 255, src/texture.cpp:16).  img_cols: (w, 4) colours of the image's columns; layer_cols_u8: list of (w, 4) uint8 (or float: a guide used as it is --
    with the image itself as the one guide this is the plain bilateral of bialteral.comp / bialteral_linear.comp away from the row ends).
    Returns (w, 4) float64, valid for columns at least `radius` away from the left / right border (and rows likewise from top / bottom)."""
    I = np.asarray(img_cols, np.float64)
    w = I.shape[0]
    es = np.exp(-0.5 * (np.arange(-radius, radius + 1, dtype=np.float64) / sigma_s) ** 2)
    num, den = np.zeros((w, 4)), np.zeros(w)
    for lc in layer_cols_u8:
        lc = np.asarray(lc)
        if lc.dtype == np.uint8:
            L = (lc.astype(np.float32)[:, :3] / np.float32(255.0)).astype(np.float64)       # the texel the shader fetches: c/255 rounded to fp32
        else:
            L = lc.astype(np.float64)[:, :3]           # a float guide: the PLAIN bilateral (bialteral.comp:29-73) is the case guide == image
        for x in range(radius, w - radius):
            d2 = ((L[x] - L[x - radius:x + radius + 1]) ** 2).sum(1)
            wt = es * np.exp(-0.5 * d2 / (sigma_c * sigma_c))
            num[x] += wt @ I[x - radius:x + radius + 1]
            den[x] += wt.sum()
    out = np.zeros((w, 4))
    ok = den > 0
    out[ok] = num[ok] / den[ok, None]
    return out


def nlm_columns_known_answer(target_cols, hparam, search, patch, neighbour_cols=None):
    """nlm_step_edge_known_answer for ANY column profile: frames whose colours depend on the column only.  The derivation is the same
    (nonlocal.comp:28-63: rows are identical, so the PW rows of a patch contribute PW times the one-row distance and the SW search rows SW
    times the same weight):
        d_f(x, sx) = PW * sum_{i in [patch)} |T(x+i) - N_f(x+sx+i)|^2_rgb,    w = exp(-d / h^2),
        out(x) = sum_f sum_sx SW w N_f(x+sx) / sum_f (0.001 + sum_sx SW w).
    target_cols: (w, 4); neighbour_cols: list of (w, 4) (default: the target itself).  Returns (w, 4) float64, valid for columns at least
    max(|search|) + max(|patch|) away from the left / right border."""
    T = np.asarray(target_cols, np.float64)
    nbs = [T] if neighbour_cols is None else [np.asarray(n, np.float64) for n in neighbour_cols]
    w = T.shape[0]
    slo, shi = search
    plo, phi = patch
    PW, SW = phi - plo, shi - slo
    m = max(-slo, shi - 1) + max(-plo, phi - 1)
    out = np.zeros((w, 4))
    xs = np.arange(m, w - m)
    num, den = np.zeros((len(xs), 4)), np.zeros(len(xs))
    for N in nbs:
        den += 0.001
        for sx in range(slo, shi):
            d = np.zeros(len(xs))
            for i in range(plo, phi):
                d += ((T[xs + i, :3] - N[xs + sx + i, :3]) ** 2).sum(1)
            wt = SW * np.exp(-(PW * d) / (hparam * hparam))
            num += wt[:, None] * N[xs + sx]
            den += wt
    out[xs] = num / den[:, None]
    return out


def nlm_additive_known_answer(f, g, hparam, search, patch, neighbours=None):
    """Known answers of shaders/nonlocal.comp for frames that vary in BOTH axes: colour(x, y) = f(x) + g(y) per channel (alpha 1).  Worked
    out by hand: with dF_i = f(x+i) - f'(x+sx+i) and dG_j = g(y+j) - g'(y+sy+j) the patch distance (:42-52, a plain sum over the PW x PW
    patch) is
        d = sum_ch [ PW sum_i dF_i^2 + PW sum_j dG_j^2 + 2 (sum_i dF_i)(sum_j dG_j) ],
    so it needs 1-D sums only; then w = exp(-d / h^2) (:55) and out = sum_f sum_s w N_f(p + s) / sum_f (0.001 + sum_s w) (:32,56-57,61-62,
    normalize.comp:42) with N_f(x, y) = f'(x) + g'(y).  No image loops, no box filters -- unlike the oracle, the NumPy restatement and the
    float64 checker.  f: (w, 3), g: (h, 3); neighbours: list of (f', g') (default: the frame itself).  Returns (h, w, 4) float64, NaN where
    the window or the patch would leave the image (margin max(|search|) + max(|patch|))."""
    f, g = np.asarray(f, np.float64), np.asarray(g, np.float64)
    nbs = [(f, g)] if neighbours is None else [(np.asarray(a, np.float64), np.asarray(b, np.float64)) for a, b in neighbours]
    w, h = f.shape[0], g.shape[0]
    slo, shi = search
    plo, phi = patch
    PW, SW = phi - plo, shi - slo
    m = max(-slo, shi - 1) + max(-plo, phi - 1)
    xs, ys = np.arange(m, w - m), np.arange(m, h - m)
    ss, pp = np.arange(slo, shi), np.arange(plo, phi)
    num = np.zeros((len(ys), len(xs), 4))
    den = np.zeros((len(ys), len(xs)))

    def sums(t, n, pos):                                   # -> sum_i d^2 over channels [pos, s], sum_i d per channel [pos, s, ch]
        d = t[pos[:, None, None] + pp[None, None, :]] - n[pos[:, None, None] + ss[None, :, None] + pp[None, None, :]]     # [pos, s, i, ch]
        return (d ** 2).sum((2, 3)), d.sum(2)
    for fn, gn in nbs:
        A2, A1 = sums(f, fn, xs)
        B2, B1 = sums(g, gn, ys)
        d = PW * A2[None, :, None, :] + PW * B2[:, None, :, None] + 2.0 * np.einsum("xsc,ytc->yxts", A1, B1)               # [y, x, sy, sx]
        wt = np.exp(-d / (hparam * hparam))
        den += 0.001 + wt.sum((2, 3))
        cx = fn[xs[:, None] + ss[None, :]]                 # candidate's f'(x + sx): [x, sx, ch]
        cy = gn[ys[:, None] + ss[None, :]]                 # candidate's g'(y + sy): [y, sy, ch]
        num[..., :3] += np.einsum("yxts,xsc->yxc", wt, cx) + np.einsum("yxts,ytc->yxc", wt, cy)
        num[..., 3] += wt.sum((2, 3))
    out = np.full((h, w, 4), np.nan)
    out[m:h - m, m:w - m] = num / den[..., None]
    return out
