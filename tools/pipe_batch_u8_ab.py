import os, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
code = r'''
import os, sys, time, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import image_denoising_filter_amd as mid, bench
dev = torch.device("cuda", 0); ctx = mid.Context(0)
fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
lf = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in fr]
W, H, N = 1920, 1080, 64
pin = mid.PinnedFrames(ctx, lf); hin = [pin.ptrs[i % 16] for i in range(N)]; hout = mid.PinnedFrames(ctx, N, W * H * 4)
def call(k):
    t0 = time.perf_counter()
    ctx.sequence_nlm_pinned(hin, hout.ptrs, W, H, mid.FMT_RGBA8, k=k, overlap=True, out_u8=True, **mid.NLM_BENCH)
    return (time.perf_counter() - t0) * 1e3
res = []
for k in (0, 2):
    call(k); call(k)
    ws = sorted(call(k) for _ in range(8 if k == 0 else 3))
    h = hashlib.sha256()
    for i in range(N): h.update(hout.array(i, (H, W, 4), np.uint8).tobytes())
    res.append("k=%d median %.2f ms = %.0f out-Mpx/s (min %.2f max %.2f) sha %s" % (k, ws[len(ws)//2], N*W*H/1e3/ws[len(ws)//2], ws[0], ws[-1], h.hexdigest()[:10]))
print("AB " + " | ".join(res), flush=True)
'''
libs = [""] + sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib: env["MID_LIB_PATH"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(f"round {rnd} {os.path.basename(lib) or 'shipped (B=1)':22s} {line[0][3:] if line else 'FAILED ' + r.stderr[-600:]}", flush=True)
