"""NLM launch time against the number of workgroup rounds (development aid): frames of 1920 x (32*k) pixels have
34*k tiles; 512 workgroups are resident at a time.  Fits t = a + b * rounds."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
W = 1920
rows = []
for k in (1, 4, 8, 15, 30, 45, 60, 120, 240):
    H = 32 * k
    fr = torch.rand((H, W, 4), device=dev) * 2
    out = torch.empty((H, W, 4), device=dev)
    def run(n):
        tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
        for _ in range(n): ctx.nlm_temporal_dev([fr.data_ptr()], [out.data_ptr()], W, H, 0.5, (-10, 11), (-3, 4), 0, 0, 1, 0, s)
        tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
    run(3); t = min(run(10) for _ in range(3))
    wgs = 34 * k
    rows.append((wgs / 512.0, t))
    print("k=%3d  %5d workgroups = %6.2f rounds  %.4f ms  (%.4f ms per round)" % (k, wgs, wgs / 512.0, t, t / (wgs / 512.0)))
x = np.array([r for r, _ in rows if r >= 1.9]); y = np.array([t for r, t in rows if r >= 1.9])
b, a = np.polyfit(x, y, 1)
print("fit over launches of >= 2 rounds: t = %.4f ms + %.4f ms * rounds" % (a, b))
