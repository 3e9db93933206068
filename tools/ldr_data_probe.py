import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import image_denoising_filter_amd as mid, bench
ctx = mid.Context(0)
dev = torch.device("cuda", 0)
fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
rng = np.random.default_rng(0)
sets = {
 "bench ldr (f*64 clipped, mostly 255)": [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in fr],
 "bench ldr scaled (f*24 clipped)": [np.clip(f * 24.0, 0, 255).astype(np.uint8) for f in fr],
 "uniform noise u8": [rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8) for _ in range(16)],
 "constant 128": [np.full((1080, 1920, 4), 128, np.uint8) for _ in range(16)],
}
for name, lf in sets.items():
    lf64 = lf * 4
    ctx.sequence_nlm(lf[:2], k=0, out_u8=True, **mid.NLM_BENCH)
    for rep in range(2):
        _, (wall, kern, copy) = ctx.sequence_nlm(lf64, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
    print(f"{name:40s} 64 frames: wall {wall:6.2f} ms kernel-sum {kern:6.2f} -> {64*1920*1080/wall/1e3:5.0f} Mpx/s", flush=True)
