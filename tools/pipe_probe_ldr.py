"""Pipeline throughput for RGBA32F and for RGBA8 in / RGBA8 out (16 x 1080p, k=0).  The library pipelines over two kernel streams (1/2/3/4 were compared with a development switch: 2385/2575/1892/1652)."""
import os, sys, subprocess
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
n = 16
hdr = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(n)]
ldr = [np.clip(f * 64, 0, 255).astype(np.uint8) for f in hdr]
for name, fr, u8 in (("hdr", hdr, False), ("ldr", ldr, True)):
    ctx.sequence_nlm(fr[:2], k=0, out_u8=u8, **mid.NLM_BENCH)
    for rep in range(2):
        outs, (wall, kern, copy) = ctx.sequence_nlm(fr, k=0, overlap=True, out_u8=u8, **mid.NLM_BENCH)
    print(f"{name}: wall {wall:.2f} ms kernel-sum {kern:.2f} copy-sum {copy:.2f} -> {n*1920*1080/wall/1e3:.0f} Mpx/s", flush=True)
'''
subprocess.run([sys.executable, "-c", code], check=True)
