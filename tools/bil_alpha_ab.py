"""A/B of library builds on the bilateral kernels incl. the layer modes, with checksums (round 6, LABNOTES R6.7: opaque-alpha fast path).
   python tools/bil_alpha_ab.py build/abl/libmi_bil_a1.so
Fresh process per library, alternated over three rounds; per configuration 7 timings of 30 launches, median and min; sha of every output.
Inputs: the bench's 1080p HDR frame (alpha = 1 everywhere) and the same frame with alpha = 0.5 in one corner tile (general path there)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, hashlib; sys.path.insert(0, sys.argv[1])
import numpy as np, torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
fr = bench.synth_frames(2, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
fr2 = fr[0].clone(); fr2[:40, :40, 3] = 0.5
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
rng = np.random.default_rng(7)
layers = [ctx.upload(rng.integers(0, 256, (bench.H, bench.W, 4), dtype=np.uint8)) for _ in range(4)]
import ctypes
def bil(img, R, lay): ctx.bilateral_dev(img.data_ptr(), out.data_ptr(), bench.W, bench.H, R, 2.0, 0.2, lay, 0, s)
def lay4(img, R):
    p = mid._lib.BilateralParams(bench.W, bench.H, 2.0, 0.2, R, 0, 0)
    tbl = (ctypes.c_void_p * 4)(*[l.ptr for l in layers])
    assert mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(p), img.data_ptr(), tbl, 4, out.data_ptr(), s) == 0
def run(fn, n=30):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): fn()
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
def sha(): torch.cuda.synchronize(); return hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:8]
for _ in range(3): run(lambda: bil(fr[0], 8, 1))
res = []
for name, fn in (("r8 linear", lambda: bil(fr[0], 8, 1)), ("r8 texture", lambda: bil(fr[0], 8, 0)), ("r4", lambda: bil(fr[0], 4, 1)), ("r10", lambda: bil(fr[0], 10, 0)),
                 ("r20", lambda: bil(fr[0], 20, 0)), ("layers4 r8", lambda: lay4(fr[0], 8)), ("layers4 r4", lambda: lay4(fr[0], 4)), ("layers4 r10", lambda: lay4(fr[0], 10)), ("layers4 r8 alpha0.5 corner", lambda: lay4(fr2, 8)), ("r8 alpha0.5 corner", lambda: bil(fr2, 8, 1)), ("r8 linear", lambda: bil(fr[0], 8, 1))):
    t = sorted(run(fn) for _ in range(7)); fn()
    res.append("%s %.4f (min %.4f) %s" % (name, t[3], t[0], sha()))
print("AB " + " | ".join(res), flush=True)
'''
libs = [""] + sys.argv[1:]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib: env["MID_LIB_PATH"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=400)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(f"round {rnd} {os.path.basename(lib) or 'shipped':20s} {line[0][3:] if line else 'FAILED ' + r.stderr[-800:]}", flush=True)
