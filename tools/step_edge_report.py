"""Observed error of the NLM kernels against the hand-derived step-edge known answers (tests/np_reference.py::nlm_step_edge_known_answer):
every interior pixel of 1080p frames; appended to profiles/r06_parity_report.txt.  LABNOTES R6.12."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_denoising_filter_amd as mid
from conftest import rel_err
from np_reference import nlm_step_edge_known_answer
H, W, m = 1080, 1920, 14
ctx = mid.Context(0)
A, B = np.float32([0.30, 0.50, 0.20, 1.0]), np.float32([0.38, 0.44, 0.26, 1.0])
cols = [(A, B), (np.float32([0.33, 0.47, 0.22, 1.0]), np.float32([0.36, 0.46, 0.21, 1.0])), (np.float32([0.27, 0.52, 0.25, 1.0]), np.float32([0.41, 0.40, 0.24, 1.0]))]
print("NLM kernels vs hand-derived step-edge known answers, 1920x1080, every interior pixel; max |got - want| / max(1, |want|)")
for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
    xe, ye = 1003, 517
    img = np.empty((H, W, 4), np.float32)
    for hp in (0.5, 0.2):
        img[:, :xe], img[:, xe:] = A, B
        want = nlm_step_edge_known_answer(W, xe, A, B, hp, search, patch)
        got = ctx.nlm_temporal([img], k=0, hparam=hp, search=search, patch=patch)[0]
        print(f"  search {search} patch {patch} h={hp}: vertical edge at column {xe}: {rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (H - 2 * m, W - 2 * m, 4))):.3e}"
              f"   (edge pixel: want {want[xe - 1, 0]:.6f}, A {A[0]:.2f}, B {B[0]:.2f})")
    img[:ye], img[ye:] = A, B
    want = nlm_step_edge_known_answer(H, ye, A, B, 0.5, search, patch)
    got = ctx.nlm_temporal([img], k=0, hparam=0.5, search=search, patch=patch)[0]
    print(f"  search {search} patch {patch} h=0.5: horizontal edge at row {ye}: {rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m, None, :], (H - 2 * m, W - 2 * m, 4))):.3e}")
    frames = []
    for a, b in cols:
        f = np.empty((H, W, 4), np.float32); f[:, :xe], f[:, xe:] = a, b; frames.append(f)
    want = nlm_step_edge_known_answer(W, xe, cols[1][0], cols[1][1], 0.5, search, patch, neighbours=cols)
    got = ctx.nlm_temporal(frames, k=1, first=1, count=1, hparam=0.5, search=search, patch=patch)[0]
    print(f"  search {search} patch {patch} h=0.5: temporal k=1 over three step-edge frames: {rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (H - 2 * m, W - 2 * m, 4))):.3e}")

# ---- layer-guided bilateral, configs[3]'s shape, layers that differ from the image (tests/np_reference.py::bilateral_layers_columns_known_answer)
from np_reference import bilateral_layers_columns_known_answer
rng = np.random.default_rng(303)
print("layer-guided bilateral (4 RGBA8 layers != image, fused launch) vs the hand-derived column-only known answers, 1920x1080, every interior pixel")
for R in (8, 4, 20):
    walk = lambda lo, hi, step: np.clip(np.cumsum(rng.normal(0, step, (W, 3)), 0) + rng.uniform(lo, hi, 3), lo, hi)
    img_cols = np.concatenate([walk(0.0, 3.0, 0.15), np.ones((W, 1))], 1).astype(np.float32)
    layer_cols = [np.concatenate([walk(0, 255, 12.0), np.full((W, 1), 255.0)], 1).astype(np.uint8) for _ in range(4)]
    img = np.ascontiguousarray(np.broadcast_to(img_cols, (H, W, 4)))
    layers = [np.ascontiguousarray(np.broadcast_to(lc, (H, W, 4))) for lc in layer_cols]
    want = bilateral_layers_columns_known_answer(img_cols, layer_cols, R, 2.0, 0.2)
    got = ctx.bilateral_layers(img, layers, R, 2.0, 0.2)
    print(f"  r = {R}: {rel_err(got[R:-R, R:-R], np.broadcast_to(want[R:-R], (H - 2 * R, W - 2 * R, 4))):.3e}   (largest shift of a pixel against the image: {np.abs(want[R:-R, :3] - img_cols[R:-R, :3]).max():.3f})")

# ---- NLM, general column / row profiles, temporal k = 2 (configs[4]'s window): tests/np_reference.py::nlm_columns_known_answer
from np_reference import nlm_columns_known_answer
print("NLM temporal k=2 over five frames with independent random-walk profiles vs the hand-derived closed form, 1920x1080, every interior pixel")
rng = np.random.default_rng(41)
walk = lambda n: np.clip(np.cumsum(rng.normal(0, 0.02, (n, 3)), 0) + rng.uniform(0.2, 0.8, 3), 0, 2)
fr_cols = [np.concatenate([walk(W), np.ones((W, 1))], 1).astype(np.float32) for _ in range(5)]
frames = [np.ascontiguousarray(np.broadcast_to(c, (H, W, 4))) for c in fr_cols]
rows = [c[:H] for c in fr_cols]
frames_t = [np.ascontiguousarray(np.broadcast_to(r[:, None, :], (H, W, 4))) for r in rows]
for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
    want = nlm_columns_known_answer(fr_cols[2], 0.5, search, patch, neighbour_cols=fr_cols)
    got = ctx.nlm_temporal(frames, k=2, first=2, count=1, hparam=0.5, search=search, patch=patch)[0]
    want_t = nlm_columns_known_answer(rows[2], 0.5, search, patch, neighbour_cols=rows)
    got_t = ctx.nlm_temporal(frames_t, k=2, first=2, count=1, hparam=0.5, search=search, patch=patch)[0]
    print(f"  search {search} patch {patch}: column profiles {rel_err(got[m:-m, m:-m], np.broadcast_to(want[m:-m], (H - 2 * m, W - 2 * m, 4))):.3e}, "
          f"row profiles {rel_err(got_t[m:-m, m:-m], np.broadcast_to(want_t[m:-m, None, :], (H - 2 * m, W - 2 * m, 4))):.3e}"
          f"   (largest shift of a pixel: {np.abs(want[m:-m, :3] - fr_cols[2][m:-m, :3]).max():.3f})")

# ---- NLM on frames that vary in both axes, colour(x, y) = f(x) + g(y): tests/np_reference.py::nlm_additive_known_answer
from conftest import additive_frames
from np_reference import nlm_additive_known_answer
print("NLM on additive frames f(x) + g(y), 360 x 480 (12 x 9 tiles), temporal k=1 over three frames / single frame, every pixel whose window stays inside")
rng = np.random.default_rng(19)
frs = additive_frames(rng, 360, 480, 3)
for search, patch in (((-7, 7), (-3, 3)), ((-10, 11), (-3, 4))):
    mm = max(-search[0], search[1] - 1) + max(-patch[0], patch[1] - 1)
    want = nlm_additive_known_answer(frs[1][0], frs[1][1], 0.5, search, patch, neighbours=[(a, b) for a, b, _ in frs])
    got = ctx.nlm_temporal([x[2] for x in frs], k=1, first=1, count=1, hparam=0.5, search=search, patch=patch)[0]
    want1 = nlm_additive_known_answer(frs[0][0], frs[0][1], 0.5, search, patch)
    got1 = ctx.nlm_temporal([frs[0][2]], k=0, hparam=0.5, search=search, patch=patch)[0]
    print(f"  search {search} patch {patch}: temporal {rel_err(got[mm:-mm, mm:-mm], want[mm:-mm, mm:-mm]):.3e}, single frame {rel_err(got1[mm:-mm, mm:-mm], want1[mm:-mm, mm:-mm]):.3e}"
          f"   (largest shift of a pixel: {np.abs(want1[mm:-mm, mm:-mm, :3] - frs[0][2][mm:-mm, mm:-mm, :3]).max():.3f})")
