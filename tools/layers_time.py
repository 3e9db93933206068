"""Layer-guided bilateral (BASELINE configs[3]: 1080p RGBA32F, 4 RGBA8 guide layers, r = 8): fused launch, ms per frame.
   python tools/layers_time.py [lib.so ...]   (fresh process per library, MID_LIB_PATH)"""
import os, subprocess, sys
code = r'''
import sys, os, ctypes; sys.path.insert(0, os.getcwd())
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
frames = bench.synth_frames(4, 100, dev); out = torch.empty((bench.H, bench.W, 4), device=dev)
lay = [(f[..., :4].clamp(0, 1) * 255).to(torch.uint8).contiguous() for f in frames[:4]]
tbl = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in lay])
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def run(n, R, L):
    bp = mid.BilateralParams(bench.W, bench.H, 2.0, 0.2, R, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n):
        rc = mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), frames[0].data_ptr(), tbl, L, out.data_ptr(), s)
        assert rc == 0, mid.lib.mid_last_error()
    tm.tock(0, s); torch.cuda.synchronize(); v = tm.ms()[0] / n; tm.close(); return v
run(3, 8, 4)
med = lambda R, L: sorted(run(10, R, L) for _ in range(5))[2]
print("layers fused: " + " | ".join("r%d L%d %.4f ms" % (R, L, med(R, L)) for R, L in ((8, 4), (8, 1), (4, 4), (10, 4))), "| checksum %.6f" % float(out.double().sum()))
'''
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib) or 'shipped':26s} {r.stdout.strip() or r.stderr[-600:]}", flush=True)
