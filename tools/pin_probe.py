"""Host-memory kinds through the frame pipeline and the plain copies, timed (LABNOTES R5.1).
16 x 1080p RGBA32F through mid_sequence_nlm: page-locked buffers; pageable arrays (bounced inside the library,
csrc/hostcopy.cpp); arrays registered in place.  Plus one 33 MB frame through mid_memcpy_h2d / _d2h, pinned vs pageable."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
n = 16
frames = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(n)]
ctx.sequence_nlm(frames[:2], k=0, pinned=False, pinned_out=False, **mid.NLM_BENCH)
for name, kw in (("page-locked in and out", dict()), ("pageable sources, page-locked outputs", dict(pinned=False)),
                 ("pageable sources and outputs", dict(pinned=False, pinned_out=False))):
    best = 1e9
    for _ in range(3):
        outs, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, **kw, **mid.NLM_BENCH)
        best = min(best, wall)
    print("%-40s pipeline wall %.1f ms = %.0f Mpixel/s" % (name, best, n * 2.0736 / best * 1e3))
t = time.perf_counter()
for f in frames: assert mid.lib.mid_host_register(ctx.handle, f.ctypes.data, f.nbytes) == 0
treg = time.perf_counter() - t
outs, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, pinned=False, **mid.NLM_BENCH)
print("registered in place: register %.1f ms (%.1f GB/s), pipeline wall %.1f ms" % (treg * 1e3, n * frames[0].nbytes / treg / 1e9, wall))
t = time.perf_counter()
for f in frames: mid.lib.mid_host_unregister(ctx.handle, f.ctypes.data)
print("unregister %.1f ms" % ((time.perf_counter() - t) * 1e3))
# one frame, plain copies
f = frames[0]; nb = f.nbytes
dev = ctx.alloc(nb)
pin = mid.PinnedFrames(ctx, [f])
out = np.empty_like(f)
def clock(fn, reps=10):
    fn(); ctx.sync()
    t = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return (time.perf_counter() - t) / reps
for name, h2d, d2h in (("page-locked", lambda: mid.lib.mid_memcpy_h2d(ctx.handle, dev.ptr, pin.ptrs[0], nb, None), lambda: mid.lib.mid_memcpy_d2h(ctx.handle, pin.ptrs[0], dev.ptr, nb, None)),
                       ("pageable", lambda: mid.lib.mid_memcpy_h2d(ctx.handle, dev.ptr, f.ctypes.data, nb, None), lambda: mid.lib.mid_memcpy_d2h(ctx.handle, out.ctypes.data, dev.ptr, nb, None))):
    a, b = clock(h2d), clock(d2h)
    print("%-12s 33 MB frame: h2d %.2f ms (%.1f GB/s), d2h %.2f ms (%.1f GB/s)" % (name, a * 1e3, nb / a / 1e9, b * 1e3, nb / b / 1e9))
t = time.perf_counter(); ctypes_copy = np.copyto(out, f); e = time.perf_counter() - t
print("host memcpy of one frame (np.copyto): %.2f ms (%.1f GB/s)" % (e * 1e3, nb / e / 1e9))
