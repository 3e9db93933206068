"""Outputs per launch in the frame pipeline (B = 1 shipped; build alternatives with
`python tools/build_alt.py pipe_b2 pipeline.cpp -DMID_PIPE_B=2`): 64 x 1080p, k=0, RGBA32F and RGBA8 in/out, five passes each
(first dropped), median and spread, one fresh process per library.   python tools/pipe_batch_ab.py [lib.so ...]"""
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
base = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(16)]
hdr = [base[i % 16] for i in range(64)]
ldr8 = [np.clip(f * 64, 0, 255).astype(np.uint8) for f in base]
ldr = [ldr8[i % 16] for i in range(64)]
for name, fr, u8 in (("hdr", hdr, False), ("ldr", ldr, True)):
    ctx.sequence_nlm(fr[:2], k=0, out_u8=u8, **mid.NLM_BENCH)
    rows = [ctx.sequence_nlm(fr, k=0, overlap=True, out_u8=u8, **mid.NLM_BENCH)[1] for _ in range(6)][1:]
    walls = sorted(r[0] for r in rows)
    mp = lambda w: 64 * 1920 * 1080 / w / 1e3
    print(f"{sys.argv[1]:24s} {name}: median {mp(walls[2]):5.0f} Mpx/s  [{mp(walls[-1]):5.0f} .. {mp(walls[0]):5.0f}]  kernel-sum {sorted(r[1] for r in rows)[2]:6.1f} ms", flush=True)
'''
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code, os.path.basename(lib) or "shipped (B=1)"], env=env, check=True)
