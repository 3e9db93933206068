"""Fixed launch mix for rocprofv3 (kernel-trace/stats and PMC passes).  Usage on the GPU box:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python tools/profile_kernels.py
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... --output-format csv -d ... -- python tools/profile_kernels.py
The same frames and parameters as bench.py's timed region (NLM 21x21/7x7).  Since round 5 every NLM launch of the mix has
bench.py's shape -- 31 output frames = 69.99 (21x21), 67.9 (14x14) rounds of workgroups on the chip's 512 slots -- so that the
counters describe the KERNELS and not the tail of a short launch: round 4's mix launched 8 frames (18.06 rounds -> 19) and, for the
temporal kernel, 4 outputs (9.03 rounds -> 10, i.e. 10 % of its cycles idle), which is most of what read as "0.76" (LABNOTES R5.4)."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_denoising_filter_amd as mid
import bench

torch.cuda.set_device(0)
ctx = mid.Context(0)
dev = torch.device("cuda", 0)
F = 31                                     # outputs per NLM launch (bench.py's default)
frames = bench.synth_frames(F + 4, 100, dev)   # + 2 halo frames on either side for the temporal launch
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
fp, op = [f.data_ptr() for f in frames], [o.data_ptr() for o in outs]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
W, H = bench.W, bench.H
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(reps):
    ctx.nlm_temporal_dev(fp[:F], op, W, H, 0.5, (-10, 11), (-3, 4), 0, 0, F, 0, s)
for _ in range(reps):
    ctx.nlm_temporal_dev(fp[:F], op, W, H, 0.5, (-7, 7), (-3, 3), 0, 0, F, 0, s)
for _ in range(reps):
    ctx.nlm_temporal_dev(fp, op, W, H, 0.5, (-10, 11), (-3, 4), 2, 2, F, 0, s)      # temporal k=2: outputs 2..32, every window 5 frames
for lay in (0, 1):
    for _ in range(reps):
        ctx.bilateral_dev(fp[0], op[0], W, H, 8, 2.0, 0.2, lay, 0, s)
for _ in range(reps):
    ctx.bilateral_dev(fp[0], op[0], W, H, 20, 2.0, 0.2, 0, 0, s)
import ctypes
lay = [(f[..., :4].clamp(0, 1) * 255).to(torch.uint8).contiguous() for f in frames[:4]]
tbl = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in lay])
bp = mid.BilateralParams(W, H, 2.0, 0.2, 8, 0, 0)
wbuf = torch.rand((H, W, 8), device=dev) + 0.5
u8 = torch.empty((H, W, 4), device=dev, dtype=torch.uint8)
zp = mid.NormalizeParams(W, H)
for _ in range(reps):
    mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), fp[0], tbl, 4, op[0], s)
    mid.lib.mid_normalize(ctx.handle, ctypes.byref(zp), wbuf.data_ptr(), op[1], s)
    mid.lib.mid_pack_u8(ctx.handle, op[1], W * H * 4, u8.data_ptr(), s)
    mid.lib.mid_unpack_u8(ctx.handle, u8.data_ptr(), W * H * 4, 0, op[2], s)
torch.cuda.synchronize()
print("done")
