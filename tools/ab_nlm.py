"""NLM timing in a fresh process per argument: 8-frame launches, single-frame launches, temporal k=2, plus a checksum line.
Used through tools/ab_nlm_libs.py to compare library BUILDS (MID_LIB_PATH).  Until round 3 the argument selected a tile
variant of a `make TUNING=1` build (MID_NLM_VARIANT); those variants left the product sources in round 4
(tools/experiments/README.md) -- the argument is now only a label, the environment variable is ignored by the library."""
import os, subprocess, sys
code = r'''
import sys, ctypes; sys.path.insert(0, ".")
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
F = 8
frames = bench.synth_frames(F, 100, dev); outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
fp, op = [f.data_ptr() for f in frames], [o.data_ptr() for o in outs]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, nf):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.nlm_temporal_dev(fp[:nf], op[:nf], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, nf, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
run(2, 8)
for rep in range(3):
    m8 = run(5, 8); m1 = run(10, 1)
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(3): ctx.nlm_temporal_dev(fp, op, bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 2, 0, 8, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); mt = tm.ms()[0] / 3
    print("variant", sys.argv[1], "batch8 %.3f ms %.0f Mpx/s | single %.3f ms %.0f Mpx/s | temporal k=2 8 frames %.3f ms %.0f out-Mpx/s %.0f pair-Mpx/s" % (m8, 8*bench.NPIX/m8/1e3, m1, bench.NPIX/m1/1e3, mt, 8*bench.NPIX/mt/1e3, 34*bench.NPIX/mt/1e3))
'''
code += r"""
run(1, 2); o = outs[1].double()
print("variant", sys.argv[1], "check sum %.9g  px %s  edge %s" % (o.sum().item(), [round(v, 7) for v in o[500, 700].tolist()], [round(v, 7) for v in o[1079, 1919].tolist()]))
"""
for v in sys.argv[1:]:
    env = dict(os.environ, MID_NLM_VARIANT=v)
    subprocess.run([sys.executable, "-c", code, v], env=env, check=True)
