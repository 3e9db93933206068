"""Lone-frame NLM launches with the HALF tail behind the main launch (shipped) and forked onto a lowest-priority stream (MID_NLM_TAIL_FORK=1): fresh process each, alternated."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, hashlib; sys.path.insert(0, sys.argv[1])
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
fr = bench.synth_frames(4, 100, dev); outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(4)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, nf):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.nlm_temporal_dev([f.data_ptr() for f in fr[:nf]], [o.data_ptr() for o in outs[:nf]], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, nf, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
run(10, 1)
r1 = sorted(run(10, 1) for _ in range(7)); r2 = sorted(run(6, 2) for _ in range(5))
run(1, 1); torch.cuda.synchronize()
print("AB lone frame median %.4f min %.4f ms | 2 frames %.4f ms | sha %s" % (r1[3], r1[0], r2[2], hashlib.sha256(outs[0].cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
'''
for rnd in range(3):
    for fork in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code, ROOT], env=dict(os.environ, MID_NLM_TAIL_FORK=fork), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(f"round {rnd} tail {'forked, lowest priority' if fork == '1' else 'behind the main launch '} {line[0][3:] if line else 'FAILED ' + r.stderr[-600:]}", flush=True)
