"""Where does the 21 % run-to-run spread of bench.py's `also.bilateral_layers_r8_L4_fused` come from (VERDICT r3 weak #5:
ms_min_max [0.646, 0.785])?  Candidates: the clock ramp after idle (DVFS), the exact-160-KiB fit of two workgroups' LDS, the
tail of a 2040-workgroup launch on 512 slots.  This probe separates them:
 (a) bench.py's own procedure -- 5 timings of 10 launches -- printed IN ORDER, once straight after another kernel and a
     synchronise (as in bench.py), once after 1 s of idle, once after 2 s of back-to-back launches;
 (b) 400 launches back to back, every launch bracketed by its own pair of events: per-launch times over the run, with the
     card's shader clock and board power sampled through its hwmon files every 5 ms;
 (c) the same for L = 1, 2, 3 layers (a launch is L passes over the tile: time should be linear in L if it is issue-bound).
Usage on the GPU box: python tools/layers_spread.py"""
import ctypes
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
import torch

import bench
import image_denoising_filter_amd as mid

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = mid.Context(0)
frames = bench.synth_frames(4, 100, dev)
out = torch.empty((bench.H, bench.W, 4), device=dev)
lay = [(f[..., :4].clamp(0, 1) * 255).to(torch.uint8).contiguous() for f in frames[:4]]
tbl = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in lay])
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
s = ts.cuda_stream
bp = mid.BilateralParams(bench.W, bench.H, 2.0, 0.2, 8, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)


def launch(L=4):
    assert mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), frames[0].data_ptr(), tbl, L, out.data_ptr(), s) == 0


def timing(n=10, L=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    for _ in range(n):
        launch(L)
    e1.record(ts)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def hwmon_paths():
    pr = torch.cuda.get_device_properties(0)
    try:
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
    except AttributeError:
        return None, None
    for card in sorted(glob.glob("/sys/class/drm/card*")):
        if addr in os.path.realpath(os.path.join(card, "device")):
            f = sorted(glob.glob(os.path.join(card, "device/hwmon/hwmon*/freq1_input")))
            p = sorted(glob.glob(os.path.join(card, "device/hwmon/hwmon*/power1_average"))) or \
                sorted(glob.glob(os.path.join(card, "device/hwmon/hwmon*/power1_input")))
            return (f[0] if f else None), (p[0] if p else None)
    return None, None


FREQ, POWER = hwmon_paths()


def sample():
    try:
        mhz = int(open(FREQ).read()) / 1e6 if FREQ else float("nan")
        w = int(open(POWER).read()) / 1e6 if POWER else float("nan")
        return mhz, w
    except Exception:  # noqa: BLE001
        return float("nan"), float("nan")


def nlm_burst():
    """another kernel first, like the extras that precede this one in bench.py"""
    f = bench.synth_frames(1, 7, dev)
    o = torch.empty((bench.H, bench.W, 4), device=dev)
    for _ in range(20):
        ctx.nlm_temporal_dev([f[0].data_ptr()], [o.data_ptr()], bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, 1, 0, s)
    torch.cuda.synchronize()


print(f"hwmon: freq {FREQ}  power {POWER}")
launch()
torch.cuda.synchronize()
# (a) bench.py's procedure, in order
for label, prep in (("after another kernel + sync (bench.py's situation)", nlm_burst),
                    ("after 1 s of idle", lambda: time.sleep(1.0)),
                    ("after 2 s of back-to-back launches", lambda: [timing(100) for _ in range(30)])):
    prep()
    launch()                                        # time_gpu()'s untimed first call
    torch.cuda.synchronize()
    t = [timing(10) for _ in range(5)]
    print(f"(a) {label:52s} 5 timings of 10 launches, in order: " + " ".join(f"{x:.4f}" for x in t) + f"  | min {min(t):.4f} max {max(t):.4f} ms ({(max(t) / min(t) - 1) * 100:.0f} %)")

# (b) per-launch times over a long run, clock and power beside them
time.sleep(1.0)
N = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(),) + sample())
        time.sleep(0.005)


th = threading.Thread(target=sampler)
th.start()
time.sleep(0.05)
t0 = time.perf_counter()
ev[0].record(ts)
for i in range(N):
    launch()
    ev[i + 1].record(ts)
torch.cuda.synchronize()
t1 = time.perf_counter()
time.sleep(0.05)
stop = True
th.join()
per = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print(f"(b) {N} launches back to back after 1 s idle: {sum(per):.1f} ms in all")
for a, b in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 80), (80, 160), (160, 400)):
    seg = per[a:b]
    print(f"    launches {a:3d}..{b - 1:3d}: mean {sum(seg) / len(seg):.4f} ms  min {min(seg):.4f}  max {max(seg):.4f}")
run_samples = [(t - t0, mhz, w) for t, mhz, w in samples]
print("    sclk / power over the run: " + " | ".join(f"t={t * 1e3:6.1f} ms {mhz:5.0f} MHz {w:5.0f} W" for t, mhz, w in run_samples[::max(1, len(run_samples) // 24)]))
steady = sorted(per[200:])
print(f"    steady state (launches 200..399): median {steady[len(steady) // 2]:.4f} ms, p5 {steady[len(steady) // 20]:.4f}, p95 {steady[-len(steady) // 20]:.4f}")

# (c) linear in the number of layers?
timing(50)
print("(c) steady-state ms per launch by layers: " + " | ".join(f"L={L} {sorted(timing(50, L) for _ in range(3))[1]:.4f}" for L in (1, 2, 3, 4)))
