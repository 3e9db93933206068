"""Single-frame NLM launches and the frame pipeline (which launches frame by frame) for a list of library builds, one fresh
process each:  python tools/single_frame_ab.py [lib.so ...]   ("" = shipped)"""
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import image_denoising_filter_amd as mid, bench
ctx = mid.Context(0)
dev = torch.device("cuda", 0)
fr = bench.synth_frames(16, 100, dev)
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(16)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in outs]
def run(n, nf):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.nlm_temporal_dev(fp[:nf], op[:nf], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, nf, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); v = tm.ms()[0] / n; tm.close(); return v
run(3, 16)
m1 = sorted(run(40, 1) for _ in range(5))[2]; m16 = sorted(run(5, 16) for _ in range(3))[1]
hf = [f.cpu().numpy() for f in fr]
hdr = [hf[i % 16] for i in range(64)]
l8 = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in hf]
ldr = [l8[i % 16] for i in range(64)]
res = {}
for name, seq, u8 in (("hdr64", hdr, False), ("ldr64", ldr, True), ("hdr16", hdr[:16], False), ("ldr16", ldr[:16], True)):
    ctx.sequence_nlm(seq[:2], k=0, out_u8=u8, **mid.NLM_BENCH)
    walls = sorted(ctx.sequence_nlm(seq, k=0, overlap=True, out_u8=u8, **mid.NLM_BENCH)[1][0] for _ in range(5))
    res[name] = len(seq) * bench.NPIX / walls[2] / 1e3
print(f"{sys.argv[1]:22s} single {m1:.3f} ms ({bench.NPIX/m1/1e3:5.0f} Mpx/s)  16-frame {m16:.3f} ms ({16*bench.NPIX/m16/1e3:5.0f})  | pipeline Mpx/s: " + "  ".join(f"{k} {v:5.0f}" for k, v in res.items()), flush=True)
'''
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code, os.path.basename(lib) or "shipped"], env=env, check=True)
