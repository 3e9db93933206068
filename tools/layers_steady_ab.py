"""The layer-guided bilateral (configs[3]: 4 RGBA8 layers, r = 8, fused) in bench.py's STEADY-STATE procedure -- 60 untimed launches,
then the median of 5 timings of 20 launches -- for a list of library builds, fresh process per library, the list walked twice.
   python tools/layers_steady_ab.py [lib.so ...]      ("" = the shipped library)"""
import os, subprocess, sys
code = r'''
import sys, os, ctypes; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
frames = bench.synth_frames(4, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
lay = [(f[..., :4].clamp(0, 1) * 255).to(torch.uint8).contiguous() for f in frames[:4]]
tbl = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in lay])
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
bp = mid.BilateralParams(bench.W, bench.H, 2.0, 0.2, 8, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)
def run(n):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n):
        assert mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), frames[0].data_ptr(), tbl, 4, out.data_ptr(), s) == 0
    tm.tock(0, s); torch.cuda.synchronize(); v = tm.ms()[0] / n; tm.close(); return v
run(60)
t = sorted(run(20) for _ in range(5))
print("4 layers r8 fused, steady state: median %.4f ms  [%.4f .. %.4f] | checksum %.6f" % (t[2], t[0], t[-1], float(out.double().sum())))
'''
libs = sys.argv[1:] or [""]
for lib in libs + libs:
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib) or 'shipped':26s} {r.stdout.strip() or r.stderr[-600:]}", flush=True)
