import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import image_denoising_filter_amd as mid, bench
dev = torch.device("cuda", 0); ctx = mid.Context(0)
fr = bench.synth_frames(16, 100, dev)
u8 = [torch.clamp(f * 64.0, 0, 255).to(torch.uint8).contiguous() for f in fr]
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(16)]
s = torch.cuda.Stream(); st = s.cuda_stream
W, H = bench.W, bench.H
def run(ptrs, fmt, n, reps=10):
    tm = bench.Timers(mid, ctx, 1)
    ctx.nlm_temporal_dev(ptrs[:n], [o.data_ptr() for o in outs[:n]], W, H, 0.5, (-10, 11), (-3, 4), 0, 0, n, fmt, st); torch.cuda.synchronize()
    tm.tick(0, st)
    for _ in range(reps): ctx.nlm_temporal_dev(ptrs[:n], [o.data_ptr() for o in outs[:n]], W, H, 0.5, (-10, 11), (-3, 4), 0, 0, n, fmt, st)
    tm.tock(0, st); torch.cuda.synchronize(); ms = tm.ms()[0] / reps; tm.close(); return ms
for rep in range(3):
    a = run([f.data_ptr() for f in fr], mid.FMT_RGBA32F, 16); b = run([f.data_ptr() for f in u8], mid.FMT_RGBA8, 16)
    c = run([f.data_ptr() for f in fr], mid.FMT_RGBA32F, 1, 30); d = run([f.data_ptr() for f in u8], mid.FMT_RGBA8, 1, 30)
    print("16 frames per launch: RGBA32F in %.3f ms = %.0f Mpx/s | RGBA8 in %.3f ms = %.0f Mpx/s || 1 frame: %.4f / %.4f ms = %.0f / %.0f Mpx/s" % (a, 16*W*H/1e3/a, b, 16*W*H/1e3/b, c, d, W*H/1e3/c, W*H/1e3/d), flush=True)
