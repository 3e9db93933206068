"""NLM with run-time search windows (the RTS instantiations of nlm_strip_kernel) and the generic fallback: ms per 1080p frame."""
import sys, os; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
fr = bench.synth_frames(4, 100, dev); outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(4)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in outs]
K = int(os.environ.get("MID_RT_K", "0"))      # temporal half-window (0: single-frame launches of 4 frames)
def run(n, search, patch):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.nlm_temporal_dev(fp, op, bench.W, bench.H, 0.5, search, patch, K, 0, 4, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); v = tm.ms()[0] / n / 4; tm.close(); return v
cases = [((-7, 8), (-2, 3)), ((-5, 6), (-1, 2)), ((-12, 13), (-3, 4)), ((-10, 11), (-2, 3)), ((-4, 5), (-2, 2)), ((-4, 4), (-4, 4)), ((-6, 7), (-4, 5)), ((-7, 8), (-3, 4)), ((-15, 16), (-3, 4)), ((-13, 14), (-2, 3)),
         ((-4, 5), (-1, 1)), ((-4, 5), (0, 1)), ((-7, 7), (-3, 3)), ((-10, 11), (-3, 4)),
         ((-10, 11), (-5, 6)), ((-7, 8), (-6, 7)),
         ((-5, 6), (-8, 8)),
         ((-4, 5), (-1, 3))]   # last: lopsided 4x4 patch, no strip instantiation -> nlm_generic_kernel
run(2, *cases[0])
print(("k=%d: " % K) + " | ".join("%dx%d/%dx%d %.3f ms" % (sr[1] - sr[0], sr[1] - sr[0], pt[1] - pt[0], pt[1] - pt[0], sorted(run(5, sr, pt) for _ in range(3))[1]) for sr, pt in cases))
