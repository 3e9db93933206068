import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
F = 16
frames = bench.synth_frames(F, 100, dev); outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(8)]
fp, op = [f.data_ptr() for f in frames], [o.data_ptr() for o in outs]
def timed(fn, n):
    fn(); torch.cuda.synchronize()
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): fn()
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
for k in (0, 1, 2, 3, 4):
    ms = timed(lambda: ctx.nlm_temporal_dev(fp, op, bench.W, bench.H, 0.5, (-10, 11), (-3, 4), k, 4, 8, 0, s), 3)
    pairs = 8 * (2 * k + 1)
    print(f"k={k}: 8 outputs, full windows, {pairs} pairs: {ms:.2f} ms -> {pairs*bench.NPIX/ms/1e3:.0f} pair-Mpx/s, {ms/pairs:.4f} ms/pair")
