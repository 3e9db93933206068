"""Bilateral r=8, texture vs linear addressing, interleaved and repeated, one fresh process per library build:
   python tools/ab_bil_layout.py [build/abl/libmi_<name>.so ...]      (no argument: the shipped library)
The two variants run the SAME inner loop (identical opcode stream); only the tile fill differs.  This
probe separates a real cost of the fill from code-placement effects (builds with -falign-loops=N move both)."""
import os
import subprocess
import sys

code = r'''
import sys, os; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
frames = bench.synth_frames(2, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, lay, R=8):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.bilateral_dev(frames[0].data_ptr(), out.data_ptr(), bench.W, bench.H, R, 2.0, 0.2, lay, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
run(20, 0); run(20, 1)
res = {0: [], 1: []}
for rep in range(9):
    for lay in ((0, 1) if rep % 2 == 0 else (1, 0)):
        res[lay].append(run(20, lay))
med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
print("%-34s texture %.4f ms (min %.4f)  linear %.4f ms (min %.4f)  texture/linear %.3f" %
      (sys.argv[1], med[0], min(res[0]), med[1], min(res[1]), med[0] / med[1]), flush=True)
'''
libs = sys.argv[1:] or [""]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code, os.path.basename(lib) or "shipped"], env=env, check=True)
