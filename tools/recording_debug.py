"""Development probe: a recorded [memset, nlm_accum x n, normalize] sequence submitted repeatedly, with eager runs in between."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import image_denoising_filter_amd as mid
from image_denoising_filter_amd.api import _check
from conftest import synth_hdr
lib = mid.lib
ctx = mid.Context(0)
h, w, n = 97, 141, 5
rng = np.random.default_rng(6)
fr_a = [synth_hdr(rng, h, w, 3.0) for _ in range(n)]
fr_b = [np.ascontiguousarray(f[:, ::-1]) for f in fr_a]
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
fr = [T(f) for f in fr_a]
Wb = torch.empty((h, w, 8), device="cuda"); out = torch.empty((h, w, 4), device="cuda")
p = mid.NlmParams(w, h, 0.5, -7, 7, -3, 3, mid.FMT_RGBA32F); pn = mid.NormalizeParams(w, h)
s = torch.cuda.Stream(); st = s.cuda_stream
def full(st):
    _check(lib.mid_memset(ctx.handle, Wb.data_ptr(), 0, Wb.numel() * 4, st), "memset")
    for i in range(n): _check(lib.mid_nlm_accum(ctx.handle, ctypes.byref(p), fr[2].data_ptr(), fr[i].data_ptr(), Wb.data_ptr(), st), "accum")
    _check(lib.mid_normalize(ctx.handle, ctypes.byref(pn), Wb.data_ptr(), out.data_ptr(), st), "norm")
def load(srcs):
    for f, a in zip(fr, srcs): f.copy_(T(a))
    torch.cuda.synchronize()
def eager():
    torch.cuda.synchronize(); full(st); ctx.sync(st); return out.clone(), Wb.clone()
wa, Wa = eager()
with ctx.record(st) as rec: full(st)
print("recorded", rec.info())
load(fr_b); wb, Wb_ = eager()
print("a != b:", not torch.equal(wa, wb))
for step, (srcs, want, wantW) in enumerate(((fr_a, wa, Wa), (fr_b, wb, Wb_), (fr_a, wa, Wa), (fr_a, wa, Wa))):
    load(srcs); out.zero_(); torch.cuda.synchronize()
    rec.submit(st); ctx.sync(st); torch.cuda.synchronize()
    print(step, "out ok", bool(torch.equal(out, want)), "W ok", bool(torch.equal(Wb, wantW)), "W[0,0]", Wb[0, 0, :5].tolist(), "want", wantW[0, 0, :5].tolist(),
          "out[0,0]", out[0, 0].tolist())
    if step == 1:
        e, _ = eager(); print("  eager again ok", bool(torch.equal(e, wb)))
