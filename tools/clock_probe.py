"""Samples the GPU's shader clock and power (rocm-smi) while 16-frame NLM launches run back to back: is the kernel
running at the 2.4 GHz the peak figures assume?  Usage on the GPU box: python tools/clock_probe.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_denoising_filter_amd as mid
import bench

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = mid.Context(0)
F = 16
frames = bench.synth_frames(F, 100, dev)
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(F)]
fp, op = [f.data_ptr() for f in frames], [o.data_ptr() for o in outs]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
samples, stop = [], False


def sampler():
    while not stop:
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10)
            keep = [l.strip() for l in r.stdout.splitlines() if "sclk" in l or "Power" in l or "mclk" in l or "junction" in l.lower()]
            samples.append((time.perf_counter(), keep))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), [f"rocm-smi failed: {e}"]))
        time.sleep(0.2)


print("idle:", subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-600:])
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(10):
        ctx.nlm_temporal_dev(fp, op, bench.W, bench.H, 0.5, (-10, 11), (-3, 4), 0, 0, F, 0, s)
    torch.cuda.synchronize(); n += 10
el = time.perf_counter() - t0
stop = True; th.join()
print(f"{n} launches in {el:.2f} s = {el / n * 1e3:.3f} ms per 16-frame launch = {F * bench.NPIX * n / el / 1e6:.0f} Mpixel/s")
for t, keep in samples:
    print(f"t={t - t0:5.2f}s", " | ".join(keep))
