"""Single-frame NLM launches for a kernel trace (development aid): rocprofv3 --kernel-trace --stats -- python3 tools/trace_single.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
fr = bench.synth_frames(2, 100, dev)
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(2)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
for _ in range(12):
    ctx.nlm_temporal_dev([fr[0].data_ptr()], [outs[0].data_ptr()], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, 1, 0, s)
torch.cuda.synchronize()
