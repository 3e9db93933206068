"""NLM kernel time at the reference's shipped windows ([-7,7) search, [-3,3) patch): 8-frame, 1-frame and temporal k=2
launches (development aid; alternative builds via MID_LIB_PATH)."""
import sys; sys.path.insert(0, ".")
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
F = 8
fr = bench.synth_frames(F, 100, dev)
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in outs]
def run(n, nf, k=0):
    tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
    for _ in range(n): ctx.nlm_temporal_dev(fp[:nf], op[:nf], bench.W, bench.H, 0.5, (-7, 7), (-3, 3), k, 0, nf, 0, s)
    tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
run(2, 8)
for rep in range(2):
    m8, m1, mt = run(10, 8), run(20, 1), run(4, 8, 2)
    print("ref windows: 8-frame %.3f ms (%.0f Mpx/s)  1-frame %.3f ms (%.0f)  temporal k=2 %.3f ms (%.0f out-Mpx/s)" % (
        m8, 8 * bench.NPIX / m8 / 1e3, m1, bench.NPIX / m1 / 1e3, mt, 8 * bench.NPIX / mt / 1e3))
print("checksum %.9g" % outs[3].double().sum().item())
