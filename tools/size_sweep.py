"""NLM and bilateral throughput by frame size (development aid): 720p .. 8K, one frame per launch and 4 per launch.
Round 6: once with opaque frames (alpha = 1.0: the short tile forms run away from the border) and once with random alpha (the general forms everywhere,
what this tool measured up to round 5)."""
import sys; sys.path.insert(0, ".")
import torch, image_denoising_filter_amd as mid, bench
torch.cuda.set_device(0); ctx = mid.Context(0); dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
import itertools
for opaque, (W, H) in itertools.product((True, False), ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320), (1000, 1000), (4097, 2161))):
    fr = [torch.rand((H, W, 4), device=dev) * 2 for _ in range(4)]
    if opaque:
        for f in fr: f[..., 3] = 1.0
    out = [torch.empty((H, W, 4), device=dev) for _ in range(4)]
    fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in out]
    def run(fn, n):
        fn(); torch.cuda.synchronize()
        tm = bench.Timers(mid, ctx, 1); tm.tick(0, s)
        for _ in range(n): fn()
        tm.tock(0, s); torch.cuda.synchronize(); return tm.ms()[0] / n
    n1 = run(lambda: ctx.nlm_temporal_dev(fp[:1], op[:1], W, H, 0.5, (-10, 11), (-3, 4), 0, 0, 1, 0, s), 5)
    n4 = run(lambda: ctx.nlm_temporal_dev(fp, op, W, H, 0.5, (-10, 11), (-3, 4), 0, 0, 4, 0, s), 3)
    b8 = run(lambda: ctx.bilateral_dev(fp[0], op[0], W, H, 8, 2.0, 0.2, 1, 0, s), 10)
    px = W * H / 1e3
    print(("alpha = 1   " if opaque else "alpha random") + " %5dx%-5d NLM 1 frame %8.3f ms %5.0f Mpx/s | 4 frames %8.3f ms %5.0f Mpx/s | bilateral r=8 %7.3f ms %6.0f Mpx/s" % (
        W, H, n1, px / n1, n4, 4 * px / n4, b8, px / b8))
