"""Do pinned 8.3 MB copies slow down by themselves some way into a burst?  (Round 6: the RGBA8 frame pipeline's downloads go from
0.19 ms to 0.67 ms per frame about 16-20 ms into a 64-frame call, whatever buffers they target.)  Two streams, N copies each
way, every copy bracketed by events; (a) nothing else on the device, (b) NLM launches running on a third stream meanwhile,
(c) as (b) but each copy waits for an event of the kernel stream first (the pipeline's dependency pattern).
Usage on the GPU box: python tools/copy_rate_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import image_denoising_filter_amd as mid
import bench

W, H, N = 1920, 1080, 96
nb = W * H * 4
ctx = mid.Context(0)
dev = torch.device("cuda", 0)
up = mid.PinnedFrames(ctx, 8, nb); down = mid.PinnedFrames(ctx, 8, nb)
d_up = [ctx.alloc(nb) for _ in range(4)]; d_down = [ctx.alloc(nb) for _ in range(4)]
s_up, s_down, s_k = (torch.cuda.Stream(device=dev) for _ in range(3))
fr = bench.synth_frames(8, 100, dev)
d_out = [ctx.alloc(W * H * 16) for _ in range(8)]
ip, op = [f.data_ptr() for f in fr], [d.ptr for d in d_out]


def burst(label, kernels, per_copy_mb=None, n=N):
    eu = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    ed = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t00 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t00.record(s_up)
    if kernels:
        for _ in range(kernels):
            ctx.nlm_temporal_dev(ip, op, W, H, 0.5, (-10, 11), (-3, 4), 0, 0, 8, mid.FMT_RGBA32F, stream=s_k.cuda_stream)
    for i in range(n):
        eu[i][0].record(s_up)
        assert mid.lib.mid_memcpy_h2d(ctx.handle, d_up[i % 4].ptr, up.ptrs[i % 8], nb, s_up.cuda_stream) == 0
        eu[i][1].record(s_up)
        ed[i][0].record(s_down)
        assert mid.lib.mid_memcpy_d2h(ctx.handle, down.ptrs[i % 8], d_down[i % 4].ptr, nb, s_down.cuda_stream) == 0
        ed[i][1].record(s_down)
    torch.cuda.synchronize()
    u = [a.elapsed_time(b) for a, b in eu]; d = [a.elapsed_time(b) for a, b in ed]
    tu = [t00.elapsed_time(a) for a, _ in eu]; td = [t00.elapsed_time(a) for a, _ in ed]
    print(f"== {label}")
    print("  H2D ms:", " ".join(f"{x:.2f}" for x in u))
    print("  D2H ms:", " ".join(f"{x:.2f}" for x in d))
    slow = next((i for i, x in enumerate(d) if x > 0.4), None)
    print(f"  first D2H slower than 0.4 ms: #{slow} at t = {td[slow]:.1f} ms" if slow is not None else "  no D2H slower than 0.4 ms",
          f"| burst ends at {max(tu[-1] + u[-1], td[-1] + d[-1]):.1f} ms")


burst("warm-up", 0, n=8)
burst("a. copies alone, both directions", 0)
burst("a2. copies alone again", 0)
burst("b. beside NLM launches (8 frames each, 3.6 ms) on a third stream", 12)
burst("b2. again", 12)
time.sleep(0.5)
burst("a3. copies alone after 0.5 s idle", 0)
