"""A/B of library builds on the plain bilateral (round 6, LABNOTES R6.4: LDS read-ahead in the tap loop).
   python tools/bil_readahead_ab.py build/abl/libmi_bil_ra.so [more.so ...]
Fresh process per library, libraries alternated over three rounds (shipped first); per process: warm-up, then 7 timings of 40
launches per configuration, median and min reported, and a checksum of the r = 8 output (same bits expected)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, hashlib; sys.path.insert(0, sys.argv[1])
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
frames = bench.synth_frames(2, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, R, lay):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.bilateral_dev(frames[0].data_ptr(), out.data_ptr(), bench.W, bench.H, R, 2.0, 0.2, lay, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
for _ in range(3): run(40, 8, 1)
res = []
for R, lay in ((8, 1), (8, 0), (4, 1), (8, 1)):
    t = sorted(run(40, R, lay) for _ in range(7))
    res.append("r%d %s median %.4f min %.4f ms" % (R, "linear" if lay else "texture", t[3], t[0]))
run(1, 8, 1); torch.cuda.synchronize()
print("AB " + " | ".join(res) + " | sha " + hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12], flush=True)
'''
libs = [""] + sys.argv[1:]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["MID_LIB_PATH"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(f"round {rnd} {os.path.basename(lib) or 'shipped':24s} {line[0][3:] if line else 'FAILED ' + r.stderr[-400:]}", flush=True)
