"""Interleaved A/B of the NLM bench launch between library BUILDS, loaded side by side with plain ctypes (only symbols every
build since round 1 exports: mid_ctx_create, mid_nlm_temporal, mid_stream_sync) -- for comparing a build against an older one
whose C-ABI lacks newer entry points (the package's own loader insists on the full table).
   python tools/ab_lib_raw.py libA.so libB.so [...]        31-frame launches, 5 rounds, order rotated every round
Prints per library the median / min / max ms per launch and a checksum of one output frame."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402  (frame generator and constants only)


class Nlm(ctypes.Structure):
    _fields_ = [("w", ctypes.c_int32), ("h", ctypes.c_int32), ("hp", ctypes.c_float), ("slo", ctypes.c_int32), ("shi", ctypes.c_int32),
                ("plo", ctypes.c_int32), ("phi", ctypes.c_int32), ("fmt", ctypes.c_int32)]


def main():
    paths = sys.argv[1:]
    F = int(os.environ.get("AB_FRAMES", "31"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    frames = bench.synth_frames(F, 100, dev)
    outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
    fp = (ctypes.c_void_p * F)(*[f.data_ptr() for f in frames])
    op = (ctypes.c_void_p * F)(*[o.data_ptr() for o in outs])
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    prm = Nlm(bench.W, bench.H, 0.5, -10, 11, -3, 4, 0)
    libs = []
    for p in paths:
        lib = ctypes.CDLL(os.path.abspath(p))
        h = ctypes.c_void_p()
        assert lib.mid_ctx_create(0, ctypes.byref(h)) == 0
        lib.mid_nlm_temporal.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p]
        libs.append((os.path.basename(p), lib, h))

    def run(lib, h, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts)
        for _ in range(n):
            assert lib.mid_nlm_temporal(h, ctypes.byref(prm), fp, F, 0, 0, F, op, ctypes.c_void_p(ts.cuda_stream)) == 0
        e1.record(ts)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    res = {name: [] for name, _, _ in libs}
    sums = {}
    for name, lib, h in libs:
        run(lib, h, 3)
        sums[name] = outs[F // 2].double().sum().item()
    for rnd in range(5):
        order = libs[rnd % len(libs):] + libs[:rnd % len(libs)]
        for name, lib, h in order:
            res[name].append(run(lib, h, 10))
    for name, v in res.items():
        v = sorted(v)
        print(f"{name:32s} {F}-frame launch: median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}  = {F * bench.NPIX / 1e3 / v[len(v) // 2]:.0f} Mpixel/s | checksum {sums[name]:.9g}")


if __name__ == "__main__":
    main()
