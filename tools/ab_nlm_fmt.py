"""NLM kernel time by input format (RGBA32F vs RGBA8 frames), 8-frame and 1-frame launches (development aid)."""
import sys; sys.path.insert(0, ".")
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
F = 8
f32 = bench.synth_frames(F, 100, dev)
u8 = [(f * 64).clamp(0, 255).to(torch.uint8).contiguous() for f in f32]
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
for name, fr, fmt in (("rgba32f", f32, 0), ("rgba8", u8, 1)):
    fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in outs]
    def run(n, nf):
        tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
        for _ in range(n): ctx.nlm_temporal_dev(fp[:nf], op[:nf], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, nf, fmt, s)
        tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
    run(2, 8)
    for rep in range(2):
        m8, m2, m1 = run(5, 8), run(10, 2), run(10, 1)
        print("%s: 8-frame %.3f ms (%.0f Mpx/s)  2-frame %.3f ms (%.0f)  1-frame %.3f ms (%.0f)" % (
            name, m8, 8 * bench.NPIX / m8 / 1e3, m2, 2 * bench.NPIX / m2 / 1e3, m1, bench.NPIX / m1 / 1e3))
