"""Times the bilateral kernels at r=4/8/10/20.  It was the A/B driver for tile shapes (MID_BIL_VARIANT selected
template instantiations that were removed once the winners were fixed in dispatch_radius, see LABNOTES.md, rounds 1-3 section 3.2);
to A/B a new shape, build a second library and point MID_LIB_PATH at it."""
import os, subprocess, sys
code = r'''
import sys, os; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
frames = bench.synth_frames(2, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, R, lay):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.bilateral_dev(frames[0].data_ptr(), out.data_ptr(), bench.W, bench.H, R, 2.0, 0.2, lay, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
run(3, 8, 0)
for rep in range(2):
    r = {R: run(20, R, 0) for R in (4, 8, 10, 20)}
    print("variant", sys.argv[1], " ".join("r%d %.4f ms %.0f Mpx/s |" % (R, m, bench.NPIX/m/1e3) for R, m in r.items()))
'''
for v in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code, v], env=dict(os.environ, MID_BIL_VARIANT=v), check=True)
