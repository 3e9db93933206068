import sys, os, ctypes; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
H, W, NPIX = bench.H, bench.W, bench.NPIX
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
wb = [torch.rand((H, W, 8), device=dev) + 0.5 for _ in range(6)]
ob = [torch.empty((H, W, 4), device=dev) for _ in range(8)]
ub = [torch.empty((H, W, 4), device=dev, dtype=torch.uint8) for _ in range(8)]
npar = mid.NormalizeParams(W, H)
rot = [0]
def nxt(n): rot[0] += 1; return rot[0] % n
def t(fn, n=48):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    for _ in range(n): fn()
    e1.record(ts); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(3):
    a = t(lambda: mid.lib.mid_normalize(ctx.handle, ctypes.byref(npar), wb[nxt(6)].data_ptr(), ob[nxt(8)].data_ptr(), s))
    b = t(lambda: mid.lib.mid_pack_u8(ctx.handle, wb[nxt(6)].data_ptr(), NPIX * 4, ub[nxt(8)].data_ptr(), s))
    c = t(lambda: mid.lib.mid_unpack_u8(ctx.handle, ub[nxt(8)].data_ptr(), NPIX * 4, 0, wb[nxt(6)].data_ptr(), s))
    print(f"normalize {a*1e3:.1f} us = {48*NPIX/a/1e9:.2f} TB/s | pack {b*1e3:.1f} us = {20*NPIX/b/1e9:.2f} TB/s | unpack {c*1e3:.1f} us = {20*NPIX/c/1e9:.2f} TB/s")
