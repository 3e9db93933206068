import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(n)]
for ov in (True, False, True):
    outs, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, overlap=ov, **mid.NLM_BENCH)
    print(f"overlap={ov} n={n}: wall {wall:.2f} ms  kernel {kern:.2f}  copy {copy:.2f}  -> {n*1920*1080/wall/1e3:.0f} Mpx/s", flush=True)
