"""First-light check on the GPU box: every kernel against the oracle on small frames, then
rough 1080p timings.  Development aid (lives under tests/ because it uses the oracle); the real suite is tests/ -m gpu."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
import image_denoising_filter_amd as mid

rng = np.random.default_rng(0)
ctx = mid.Context(0)
print("device:", ctx.name)

def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))

h, w = 45, 83
img = (rng.random((h, w, 4), dtype=np.float32) * 2.0).astype(np.float32)
img8 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
for R in (3, 4, 8, 10):
    for lay, orc in (("texture", oracle.bilateral_texture), ("linear", oracle.bilateral_linear)):
        print("bilateral", lay, R, rel(ctx.bilateral(img, R, layout=lay), orc(img, R)))
print("bilateral u8 tex r4", rel(ctx.bilateral(img8, 4), oracle.bilateral_texture(oracle.unpack_u8(img8), 4)))
lay8 = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(3)]
W = np.zeros((h, w, 8), np.float32)
Wg = ctx.bilateral_layers_accum(img, lay8[0], W, 4); Wo = oracle.bilateral_layers_accum(img, lay8[0], W, 4)
print("layers accum r4", rel(Wg, Wo))
Wo2 = W
for l in lay8: Wo2 = oracle.bilateral_layers_accum(img, l, Wo2, 4)
print("layers fused r4", rel(ctx.bilateral_layers(img, lay8, 4), oracle.normalize(Wo2)))
print("normalize exact", np.array_equal(ctx.normalize(Wo2), oracle.normalize(Wo2)))
for name, cfg in (("ref", mid.NLM_REFERENCE), ("bench", mid.NLM_BENCH), ("odd", dict(search=(-2, 3), patch=(-1, 2)))):
    t = (rng.random((h, w, 4), dtype=np.float32)).astype(np.float32)
    nb = np.clip(t + 0.05 * rng.standard_normal((h, w, 4)).astype(np.float32), 0, 1).astype(np.float32)
    Wg = ctx.nlm_accum(t, nb, W, 0.5, **cfg); Wo = oracle.nlm_accum(t, nb, W, 0.5, **cfg)
    print("nlm accum", name, rel(Wg, Wo))
    fr = [t, nb, np.clip(nb + 0.05 * rng.standard_normal((h, w, 4)).astype(np.float32), 0, 1).astype(np.float32)]
    og = ctx.nlm_temporal(fr, k=1, **cfg); oo = oracle.nlm_temporal(fr, k=1, **cfg)
    print("nlm temporal", name, [rel(a, b) for a, b in zip(og, oo)])
u8 = np.arange(256, dtype=np.uint8).repeat(4)
print("unpack exact", np.array_equal(ctx.unpack_u8(u8, 0), oracle.unpack_u8(u8, 0)), np.array_equal(ctx.unpack_u8(u8, 1), oracle.unpack_u8(u8, 1)))
f = (rng.random(4096, dtype=np.float32) * 1.2 - 0.1).astype(np.float32)
print("pack exact", np.array_equal(ctx.pack_u8(f), oracle.pack_u8(f)))

# rough 1080p timings
import ctypes
from image_denoising_filter_amd import lib
H, Wd = 1080, 1920
big = (rng.random((H, Wd, 4), dtype=np.float32) * 4).astype(np.float32)
d_in = ctx.upload(big); d_out = ctx.alloc(H * Wd * 16)
tm = ctypes.c_void_p(); lib.mid_timer_create(ctx.handle, ctypes.byref(tm))
def timeit(fn, n=5):
    fn(); ctx.sync()
    lib.mid_timer_tick(tm, None)
    for _ in range(n): fn()
    lib.mid_timer_tock(tm, None)
    ms = ctypes.c_float(); lib.mid_timer_ms(tm, ctypes.byref(ms))
    return ms.value / n
for R in (4, 8, 10, 20):
    for lay in (0, 1):
        ms = timeit(lambda: ctx.bilateral_dev(d_in.ptr, d_out.ptr, Wd, H, R, 2.0, 0.2, lay, 0))
        print(f"bilateral r={R} layout={lay}: {ms:.3f} ms  {H*Wd/ms/1e3:.0f} Mpx/s")
for name, cfg in (("ref", mid.NLM_REFERENCE), ("bench", mid.NLM_BENCH)):
    for nfr in (1, 8):
        outs = [ctx.alloc(H * Wd * 16) for _ in range(nfr)]
        ms = timeit(lambda: ctx.nlm_temporal_dev([d_in.ptr] * nfr, [o.ptr for o in outs], Wd, H, 0.5, cfg["search"], cfg["patch"], 0, 0, nfr, 0), n=3)
        print(f"nlm {name} batch={nfr}: {ms:.3f} ms  {nfr*H*Wd/ms/1e3:.0f} Mpx/s")
for R in (5, 6, 12, 16, 24):
    ms = timeit(lambda: ctx.bilateral_dev(d_in.ptr, d_out.ptr, Wd, H, R, 2.0, 0.2, 0, 0))
    print(f"bilateral (run-time radius kernel) r={R}: {ms:.3f} ms  {H*Wd/ms/1e3:.0f} Mpx/s")
for name, cfg in (("15x15/5x5", dict(search=(-7, 8), patch=(-2, 3))), ("11x11/3x3", dict(search=(-5, 6), patch=(-1, 2))), ("25x25/7x7", dict(search=(-12, 13), patch=(-3, 4))), ("9x9/4x4 (generic)", dict(search=(-4, 5), patch=(-2, 2)))):
    o = ctx.alloc(H * Wd * 16)
    ms = timeit(lambda: ctx.nlm_temporal_dev([d_in.ptr], [o.ptr], Wd, H, 0.5, cfg["search"], cfg["patch"], 0, 0, 1, 0), n=3)
    print(f"nlm {name}: {ms:.3f} ms  {H*Wd/ms/1e3:.0f} Mpx/s")
