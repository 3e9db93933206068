"""Is the fused temporal (MULTI) kernel slower per (pixel, neighbour frame) than the single-pair kernels?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
F = 8
frames = bench.synth_frames(F, 100, dev); outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
fp, op = [f.data_ptr() for f in frames], [o.data_ptr() for o in outs]
Wb = torch.zeros((bench.H, bench.W, 8), device=dev)
p = mid.NlmParams(bench.W, bench.H, 0.5, -10, 11, -3, 4, 0)
zp = mid.NormalizeParams(bench.W, bench.H)
def timed(fn, n):
    fn(); torch.cuda.synchronize()
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): fn()
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
def fused(): ctx.nlm_temporal_dev(fp, op, bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 2, 0, F, 0, s)
def unfused():
    for t in range(F):
        mid.lib.mid_memset(ctx.handle, Wb.data_ptr(), 0, Wb.numel() * 4, s)
        for f in range(max(0, t - 2), min(F - 1, t + 2) + 1):
            mid.lib.mid_nlm_accum(ctx.handle, ctypes.byref(p), fp[t], fp[f], Wb.data_ptr(), s)
        mid.lib.mid_normalize(ctx.handle, ctypes.byref(zp), Wb.data_ptr(), op[t], s)
def single_batch(): ctx.nlm_temporal_dev(fp, op, bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, F, 0, s)
a, b, c = timed(fused, 3), timed(unfused, 3), timed(single_batch, 5)
print(f"fused temporal k=2: {a:.2f} ms ({34*bench.NPIX/a/1e3:.0f} pair-Mpx/s) | 34 accum + 8 normalize launches: {b:.2f} ms ({34*bench.NPIX/b/1e3:.0f}) | single-frame batch: {c:.2f} ms ({8*bench.NPIX/c/1e3:.0f})")
