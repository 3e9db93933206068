import os, sys, subprocess
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
n = 16
frames = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(n)]
ctx.sequence_nlm(frames[:2], k=0, **mid.NLM_BENCH)
for k in (0, 2):
    for ov in (True, False):
        outs, (wall, kern, copy) = ctx.sequence_nlm(frames, k=k, overlap=ov, **mid.NLM_BENCH)
        print(f"k={k} overlap={ov}: wall {wall:.2f} ms kernel {kern:.2f} copy {copy:.2f} -> {n*1920*1080/wall/1e3:.0f} Mpx/s", flush=True)
'''
subprocess.run([sys.executable, "-c", code], check=True)   # (a fresh process: cold pinned-memory pools like a real caller)
