"""NLM 21x21/7x7, 1080p: ms per launch by frames per launch (1..8, 16) for a list of library builds (fresh process each, alternated twice).
Round 6: where is the crossover between the scheduling strategies (max-ILP vs iterative-ILP) of the two-form kernels?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os; sys.path.insert(0, sys.argv[1])
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
fr = bench.synth_frames(16, 100, dev); outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(16)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, nf):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.nlm_temporal_dev([f.data_ptr() for f in fr[:nf]], [o.data_ptr() for o in outs[:nf]], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, nf, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
run(10, 1); run(3, 8)
res = []
for nf in (1, 2, 3, 4, 6, 8, 16):
    t = sorted(run(max(2, 12 // nf), nf) for _ in range(5))
    res.append("%d: %.3f" % (nf, t[2]))
print("AB ms per launch by frames: " + " | ".join(res), flush=True)
'''
libs = [""] + sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib: env["MID_LIB_PATH"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(f"round {rnd} {os.path.basename(lib) or 'shipped':24s} {line[0][3:] if line else 'FAILED ' + r.stderr[-600:]}", flush=True)
