import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
n = 16
frames = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(n)]
ctx.sequence_nlm(frames[:2], k=0, pinned=False, **mid.NLM_BENCH)
t = time.perf_counter(); outs, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, pinned=False, **mid.NLM_BENCH); e = time.perf_counter() - t
print("pageable sources (outputs pinned by the wrapper): pipeline wall %.1f ms, call %.1f ms" % (wall, e * 1e3))
t = time.perf_counter()
for f in frames: assert mid.lib.mid_host_register(ctx.handle, f.ctypes.data, f.nbytes) == 0
treg = time.perf_counter() - t
t = time.perf_counter(); outs, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, pinned=False, **mid.NLM_BENCH); e = time.perf_counter() - t
print("registered in place: register %.1f ms (%.1f GB/s), pipeline wall %.1f ms" % (treg * 1e3, n * frames[0].nbytes / treg / 1e9, wall))
t = time.perf_counter()
for f in frames: mid.lib.mid_host_unregister(ctx.handle, f.ctypes.data)
print("unregister %.1f ms" % ((time.perf_counter() - t) * 1e3))
