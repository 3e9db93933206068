"""Which engine moves the frame pipeline's host<->device copies, and what does the link give?  (the
pipeline's copies showed up as `__amd_rocclr_copyBuffer` blit KERNELS in the round-3 kernel trace -- kernels that take CUs
from the NLM launches -- cause not established.)
Run it plainly, with HSA_ENABLE_SDMA=0 and with HSA_ENABLE_SDMA=1 (exported in the shell, each in a fresh process), each once
under `rocprofv3 --kernel-trace --stats` to count the blit kernels.  Prints:
  * pinned hipMemcpyAsync rates: H2D alone, D2H alone, both directions at once (16 x 33 MB frames);
  * the 16-frame and 64-frame RGBA32F pipeline and the RGBA8 one (clock around the C call, steady state)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import image_denoising_filter_amd as mid

print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"), "| HIP/ROCR env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "ROCR_", "GPU_"))})
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = mid.Context(0)
NB = bench.NPIX * 16
n = 16
up, down = mid.PinnedFrames(ctx, n, NB), mid.PinnedFrames(ctx, n, NB)
d_up, d_down = ctx.alloc(NB), ctx.alloc(NB)
s_up, s_down = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def copies(do_up, do_down):
    for i in range(n):
        if do_up:
            assert mid.lib.mid_memcpy_h2d(ctx.handle, d_up.ptr, up.ptrs[i], NB, s_up.cuda_stream) == 0
        if do_down:
            assert mid.lib.mid_memcpy_d2h(ctx.handle, down.ptrs[i], d_down.ptr, NB, s_down.cuda_stream) == 0
    ctx.sync(s_up.cuda_stream)
    ctx.sync(s_down.cuda_stream)


def rate(do_up, do_down):
    copies(do_up, do_down)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        copies(do_up, do_down)
        ts.append(time.perf_counter() - t0)
    return n * NB / sorted(ts)[1] / 1e9


print(f"pinned copies of {NB >> 20} MiB: H2D alone {rate(True, False):.1f} GB/s | D2H alone {rate(False, True):.1f} GB/s | both at once {rate(True, True):.1f} GB/s each way")

frames = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]


def pipeline(fr, out_u8, label):
    uniq = {}
    for f in fr:
        uniq.setdefault(id(f), f)
    pin = mid.PinnedFrames(ctx, list(uniq.values()))
    ptr = dict(zip(uniq.keys(), pin.ptrs))
    hin = [ptr[id(f)] for f in fr]
    h_, w_ = fr[0].shape[:2]
    hout = mid.PinnedFrames(ctx, len(fr), w_ * h_ * (4 if out_u8 else 16))
    fmt = mid.FMT_RGBA8 if fr[0].dtype == np.uint8 else mid.FMT_RGBA32F
    rows = []
    for _ in range(5):
        t0 = time.perf_counter()
        inside = ctx.sequence_nlm_pinned(hin, hout.ptrs, w_, h_, fmt, k=0, overlap=True, search=(-10, 11), patch=(-3, 4), out_u8=out_u8)
        rows.append(((time.perf_counter() - t0) * 1e3, inside))
    rows = sorted(rows[1:], key=lambda r: r[0])
    wall, (_, kern, copy) = rows[len(rows) // 2]
    print(f"pipeline {label}: {len(fr)} frames {wall:.2f} ms = {len(fr) * bench.NPIX / 1e3 / wall:.0f} Mpixel/s | kernel-time sum {kern:.1f} ms, copy-time sum {copy:.1f} ms")
    pin.free()
    hout.free()


pipeline(frames, False, "RGBA32F")
pipeline([frames[i % 16] for i in range(64)], False, "RGBA32F")
ldr = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in frames]
pipeline([ldr[i % 16] for i in range(64)], True, "RGBA8 in/out")
