"""Pass-to-pass spread of the 64-frame RGBA8 pipeline (bench.py's also.pipeline_pcie_inclusive_ldr_64): N passes in one process, per pass
the wall time, the call's kernel and copy sums and, from the call's own events (mid_pipe_last_timeline), what gated its launches.
   python tools/pipe_u8_spread_probe.py [passes]
Used for LABNOTES R6.1b: one pass in four of BENCH-like runs read 17 % slow; where does such a pass lose its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import image_denoising_filter_amd as mid
import bench
from pipeline_u8_ab import analyse
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0); ctx = mid.Context(0)
fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
lf = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in fr]
W, H, N = 1920, 1080, 64
pin = mid.PinnedFrames(ctx, lf); hin = [pin.ptrs[i % 16] for i in range(N)]; hout = mid.PinnedFrames(ctx, N, W * H * 4)
def call():
    t0 = time.perf_counter()
    t = ctx.sequence_nlm_pinned(hin, hout.ptrs, W, H, mid.FMT_RGBA8, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
    return (time.perf_counter() - t0) * 1e3, t
call(); call()
rows = []
for i in range(passes):
    if i % 10 == 5: time.sleep(0.3)                 # an idle gap now and then, like the gaps between bench.py's extras
    w, t = call()
    up, out = ctx.pipe_last_timeline()
    _, s = analyse(up, out, 0, "")
    ups = [e - s_ for _, s_, e in up]
    rows.append((w, t[1], t[2], s))
    print(f"pass {i:2d}: wall {w:6.2f} ms = {N * W * H / 1e3 / w:6.0f} Mpx/s | kernel sum {t[1]:6.2f} copy sum {t[2]:6.2f} | upload avg {s['avg_upload_ms']:.3f} max {max(ups):.3f} | "
          f"kernel avg {s['avg_kernel_ms']:.3f} busy {s['kernel_stream_busy_frac']:.2f} | idle by gate {s['kernel_stream_idle_ms_by_gate']} gated {s['launches_gated_by']}", flush=True)
ws = sorted(r[0] for r in rows)
print(f"median {ws[len(ws) // 2]:.2f} ms, min {ws[0]:.2f}, max {ws[-1]:.2f}; passes more than 3 % above the median: {sum(1 for w in ws if w > 1.03 * ws[len(ws) // 2])} of {len(ws)}")
