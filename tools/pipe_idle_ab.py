"""The frame pipeline against one batched launch (16 x 1080p, or 64 with --long; k=0; HDR = RGBA32F in/out, LDR = RGBA8
in/out).  Each line is a fresh process: the event-joined pipeline, and -- on the "plain" line -- ONE batched launch over
the same frames resident in HBM, back to back and after an idle GPU (the clock ramp that separates the 16-frame pipeline
figures from the kernel's batched rate).  (Round 2 also ran the gated chunk-launch schedule here; it measured no gain and
was removed from the library in round 3 -- last present at commit 06500d5.)"""
import os, sys, subprocess
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
n = int(os.environ.get('PROBE_FRAMES', '16'))
hdr = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(n)]
ldr = [np.clip(f * 64, 0, 255).astype(np.uint8) for f in hdr]
for name, fr, u8 in (("hdr", hdr, False), ("ldr", ldr, True)):
    ctx.sequence_nlm(fr[:2], k=0, out_u8=u8, **mid.NLM_BENCH)
    best = None
    for rep in range(3):
        outs, (wall, kern, copy) = ctx.sequence_nlm(fr, k=0, overlap=True, out_u8=u8, **mid.NLM_BENCH)
        if best is None or wall < best[0]: best = (wall, kern, copy)
    wall, kern, copy = best
    print(f"{sys.argv[1]:28s} {name}: wall {wall:6.2f} ms kernel-sum {kern:6.2f} copy-sum {copy:6.2f} -> {n*1920*1080/wall/1e3:5.0f} Mpx/s", flush=True)
    if sys.argv[1].startswith("plain"):
        import ctypes
        d_in = [ctx.upload(f) for f in fr]
        d_out = [ctx.alloc(1920 * 1080 * 16) for _ in fr]
        tm = ctypes.c_void_p(); mid.lib.mid_timer_create(ctx.handle, ctypes.byref(tm))
        ip, op = [d.ptr for d in d_in], [d.ptr for d in d_out]
        ctx.nlm_temporal_dev(ip, op, 1920, 1080, 0.5, (-10, 11), (-3, 4), 0, 0, n, 1 if u8 else 0); ctx.sync()
        mid.lib.mid_timer_tick(tm, None)
        for _ in range(3): ctx.nlm_temporal_dev(ip, op, 1920, 1080, 0.5, (-10, 11), (-3, 4), 0, 0, n, 1 if u8 else 0)
        mid.lib.mid_timer_tock(tm, None); ms = ctypes.c_float(); mid.lib.mid_timer_ms(tm, ctypes.byref(ms))
        print(f"{'plain launch, warm':28s} {name}: {ms.value / 3:6.2f} ms per {n}-frame launch (same frames, resident, back to back)", flush=True)
        import time
        for idle in (0.05, 0.5):
            ctx.sync(); time.sleep(idle)
            mid.lib.mid_timer_tick(tm, None)
            ctx.nlm_temporal_dev(ip, op, 1920, 1080, 0.5, (-10, 11), (-3, 4), 0, 0, n, 1 if u8 else 0)
            mid.lib.mid_timer_tock(tm, None); mid.lib.mid_timer_ms(tm, ctypes.byref(ms))
            print(f"{'plain launch, after idle':28s} {name}: {ms.value:6.2f} ms for one {n}-frame launch after {idle*1e3:.0f} ms of idle GPU", flush=True)
'''
runs = [("event-joined", {}), ("plain", {})]
if "--long" in sys.argv:
    os.environ["PROBE_FRAMES"] = "64"
for label, env in runs:
    subprocess.run([sys.executable, "-c", code, label], env=dict(os.environ, **env), check=True)
