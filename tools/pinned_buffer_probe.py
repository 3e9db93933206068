"""Is the slow half of the RGBA8 pipeline's downloads a property of the DESTINATION BUFFERS (where the pinned pages live) or of
the copy's position in the call?  (a) one idle-device D2H / H2D copy per pinned buffer, timed alone; (b) the NUMA node of each
buffer's pages from /proc/self/numa_maps; (c) the 64-frame u8 pipeline with the output buffers in allocation order and reversed:
does the slow half follow the buffers?  Usage on the GPU box: python tools/pinned_buffer_probe.py"""
import ctypes, os, re, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import image_denoising_filter_amd as mid
import bench

W, H, N = 1920, 1080, 64
ctx = mid.Context(0)
dev = torch.device("cuda", 0)
fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
lf = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in fr]
pin = mid.PinnedFrames(ctx, lf)
hout = mid.PinnedFrames(ctx, N, W * H * 4)
nb = W * H * 4
d = ctx.alloc(nb)
tm = ctypes.c_void_p(); mid.lib.mid_timer_create(ctx.handle, ctypes.byref(tm))


def timed(fn):
    mid.lib.mid_timer_tick(tm, None); fn(); mid.lib.mid_timer_tock(tm, None)
    ms = ctypes.c_float(); mid.lib.mid_timer_ms(tm, ctypes.byref(ms)); return ms.value


def node_of(addr):
    best = None
    for line in open("/proc/self/numa_maps"):
        a = int(line.split()[0], 16)
        if a <= addr and (best is None or a > best[0]):
            best = (a, line.strip())
    return best[1][:160] if best else "?"


print("cpus:", os.cpu_count(), "| numa nodes:", sorted(x for x in os.listdir("/sys/devices/system/node") if x.startswith("node")))
for rep in range(2):
    ms = []
    for i in range(N):
        t = timed(lambda: mid.lib.mid_memcpy_d2h(ctx.handle, hout.ptrs[i], d.ptr, nb, None))
        ms.append(t)
    print(f"idle D2H per buffer, pass {rep} (ms):", " ".join(f"{x:.3f}" for x in ms))
for i in (0, 15, 30, 31, 32, 47, 63):
    print(f"buffer {i:2d} @ {hout.ptrs[i]:#x}: {node_of(hout.ptrs[i])}")


def run(order, label):
    hin = [pin.ptrs[i % 16] for i in range(N)]
    ho = [hout.ptrs[i] for i in order]
    for _ in range(3):
        ctx.sequence_nlm_pinned(hin, ho, W, H, mid.FMT_RGBA8, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
    t0 = time.perf_counter()
    ctx.sequence_nlm_pinned(hin, ho, W, H, mid.FMT_RGBA8, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
    wall = (time.perf_counter() - t0) * 1e3
    up, out = ctx.pipe_last_timeline()
    print(f"{label}: wall {wall:.2f} ms; download ms per frame:", " ".join(f"{o[4] - o[3]:.2f}" for o in out))
    print(f"{label}: upload ms per frame:  ", " ".join(f"{u[2] - u[1]:.2f}" for u in up))


run(list(range(N)), "outputs in allocation order")
run(list(range(N - 1, -1, -1)), "outputs reversed")
run(list(range(N)), "outputs in allocation order again")
run([i % 8 for i in range(N)], "only 8 distinct output buffers, cycled")
