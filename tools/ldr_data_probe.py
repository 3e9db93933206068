"""Why does the RGBA8 pipeline figure move with image content?  For four kinds of 8-bit content this probe measures
(a) the host->host pipeline over 64 frames (what bench.py's also.pipeline_pcie_inclusive_ldr_64 reports), repeated back to
    back for ~0.5 s while the shader clock and board power are sampled (amdgpu hwmon: freq1_input / power1_average|input;
    rocm-smi as a fallback), and
(b) the kernel alone: ONE 16-frame launch over the same frames resident in HBM, back to back (no copies, no launch gaps).
The kernel's instruction stream does not depend on the data; its clock does (the chip lowers its clock under load, and how
much depends on how many bits toggle: MI355X_MICROARCH.md, 'DVFS give-back').  Usage on the GPU box: python tools/ldr_data_probe.py"""
import ctypes
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import image_denoising_filter_amd as mid
import bench

ctx = mid.Context(0)
dev = torch.device("cuda", 0)
fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
rng = np.random.default_rng(0)
sets = {
    "bench ldr (f*64 clipped)": [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in fr],
    "bench ldr scaled (f*24 clipped)": [np.clip(f * 24.0, 0, 255).astype(np.uint8) for f in fr],
    "uniform noise u8": [rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8) for _ in range(16)],
    "constant 128": [np.full((1080, 1920, 4), 128, np.uint8) for _ in range(16)],
}


def hwmon_paths():
    """hwmon files of THE card this process computes on (a box has eight; cuda:0 is whichever one the lease exposes):
    matched through the PCI address HIP reports for device 0."""
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
    """(sclk MHz, power W) or None."""
    try:
        if FREQ:
            mhz = int(open(FREQ).read()) / 1e6
            w = int(open(POWER).read()) / 1e6 if POWER else float("nan")
            return mhz, w
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        mhz = [float(l.split("(")[1].split("Mhz")[0]) for l in r.splitlines() if "sclk" in l and "(" in l]
        w = [float(l.split(":")[-1]) for l in r.splitlines() if "Power" in l and "W" in l.split(":")[-2]]
        return (mhz[0] if mhz else float("nan")), (w[0] if w else float("nan"))
    except Exception:  # noqa: BLE001
        return None


class Sampler:
    def __enter__(self):
        self.s, self.stop = [], False
        self.t = threading.Thread(target=self.loop)
        self.t.start()
        return self

    def loop(self):
        while not self.stop:
            v = sample()
            if v:
                self.s.append(v)
            time.sleep(0.01 if FREQ else 0.1)

    def __exit__(self, *a):
        self.stop = True
        self.t.join()

    def mean(self):
        if not self.s:
            return float("nan"), float("nan"), 0
        a = np.array(self.s, dtype=np.float64)
        return float(np.nanmean(a[:, 0])), float(np.nanmean(a[:, 1])), len(a)


print(f"sampling: {'hwmon ' + FREQ if FREQ else 'rocm-smi (coarse)'}", flush=True)
for name, lf in sets.items():
    lf64 = [lf[i % 16] for i in range(64)]
    ctx.sequence_nlm(lf[:2], k=0, out_u8=True, **mid.NLM_BENCH)
    ctx.sequence_nlm(lf64, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)          # first use of the larger buffers
    walls, kerns = [], []
    with Sampler() as sm:
        for rep in range(10):
            _, (wall, kern, copy) = ctx.sequence_nlm(lf64, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
            walls.append(wall); kerns.append(kern)
    mhz, watts, ns = sm.mean()
    wall = sorted(walls)[len(walls) // 2]
    print(f"{name:34s} pipeline 64 frames: wall {wall:6.2f} ms (min {min(walls):6.2f}) kernel-sum {sorted(kerns)[5]:6.2f} -> "
          f"{64 * 1920 * 1080 / wall / 1e3:5.0f} Mpx/s | sclk {mhz:6.0f} MHz, {watts:5.0f} W ({ns} samples; includes the host-side gaps between the 10 passes)", flush=True)
    # the kernel alone, resident frames, back to back
    d_in = [ctx.upload(f) for f in lf]
    d_out = [ctx.alloc(1920 * 1080 * 16) for _ in lf]
    ip, op = [d.ptr for d in d_in], [d.ptr for d in d_out]
    tm = ctypes.c_void_p()
    mid.lib.mid_timer_create(ctx.handle, ctypes.byref(tm))
    for _ in range(3):
        ctx.nlm_temporal_dev(ip, op, 1920, 1080, 0.5, (-10, 11), (-3, 4), 0, 0, 16, 1)
    ctx.sync()
    with Sampler() as sm:
        mid.lib.mid_timer_tick(tm, None)
        for _ in range(40):
            ctx.nlm_temporal_dev(ip, op, 1920, 1080, 0.5, (-10, 11), (-3, 4), 0, 0, 16, 1)
        mid.lib.mid_timer_tock(tm, None)
        ms = ctypes.c_float()
        mid.lib.mid_timer_ms(tm, ctypes.byref(ms))
    mhz, watts, ns = sm.mean()
    print(f"{'':34s} kernel alone, 16-frame launches x40: {ms.value / 40:6.3f} ms per launch = {16 * 1920 * 1080 / (ms.value / 40) / 1e3:5.0f} Mpx/s | "
          f"sclk {mhz:6.0f} MHz, {watts:5.0f} W ({ns} samples)", flush=True)
    del d_in, d_out
