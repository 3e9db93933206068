import sys, os; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
frames = bench.synth_frames(2, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, R):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.bilateral_dev(frames[0].data_ptr(), out.data_ptr(), bench.W, bench.H, R, 2.0, 0.2, 0, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); v = tm.ms()[0] / n; tm.close(); return v
run(5, 6)
print(" | ".join("r%d %.4f ms" % (R, sorted(run(20, R) for _ in range(3))[1]) for R in (3, 6, 12, 16, 24)))
