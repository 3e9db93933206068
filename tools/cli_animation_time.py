"""End-to-end wall time of `mi_denoise --animation` on N x 1080p frames (PNG and EXR), files decoded / encoded one at a time
(--io-threads 1: the behaviour up to round 5: each file uses the codec's internal parallelism, up to 16 threads) against one file per worker thread (default).
   python tools/cli_animation_time.py [frames=64]
LABNOTES R6.6: the GPU needs 33 ms (PNG) / 46 ms (EXR) for the sequence; the drop-in command's time is the codecs'."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import image_denoising_filter_amd as mid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CLI = os.path.join(ROOT, "image_denoising_filter_amd", "mi_denoise")
dev = torch.device("cuda", 0)
fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
print("host threads:", os.cpu_count(), flush=True)
for ext in ("png", "exr"):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        for i in range(n):
            a = fr[i % 16]
            mid.save_image(os.path.join(d, f"Animation01_LDR_{i:04d}.{ext}"), np.clip(a * 64.0, 0, 255).astype(np.uint8) if ext == "png" else a)
        print(f"== {n} x 1080p .{ext}: written in {time.perf_counter() - t0:.1f} s ({sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6:.0f} MB)", flush=True)
        for label, extra in (("one file at a time (--io-threads 1)", ["--io-threads", "1"]), ("one file per worker thread (default)", []), ("again", [])):
            out = os.path.join(d, "out_" + label[:3].strip())
            os.makedirs(out, exist_ok=True)
            t0 = time.perf_counter()
            r = subprocess.run([CLI, os.path.join(d, f"Animation01_LDR_0000.{ext}"), "--animation", "--temporal-k", "0", "--search", "-10,11", "--patch", "-3,4",
                                "--outdir", out] + extra, capture_output=True, text=True, cwd=d)
            wall = time.perf_counter() - t0
            if r.returncode:
                print("FAILED", r.stdout[-800:], r.stderr[-800:]); sys.exit(1)
            lines = [l.strip() for l in r.stdout.splitlines() if "decoded" in l or "end to end" in l or "encoded" in l]
            print(f"  {label}: wall {wall:.2f} s | " + " | ".join(lines), flush=True)
        if ext == "png":
            # the reference's own command line: six GPU modes over the directory (two of them load every sibling frame) + the CPU runs
            for label, extra in (("one file at a time", ["--io-threads", "1"]), ("one file per worker thread", [])):
                out = os.path.join(d, "modes_" + label[:8].replace(" ", "_"))
                os.makedirs(out, exist_ok=True)
                t0 = time.perf_counter()
                r = subprocess.run([CLI, os.path.join(d, f"Animation01_LDR_0000.{ext}"), "--outdir", out, "--cpu-threads", "16"] + extra, capture_output=True, text=True, cwd=d)
                wall = time.perf_counter() - t0
                if r.returncode:
                    print("FAILED", r.stdout[-800:], r.stderr[-800:]); sys.exit(1)
                cpu = [l.strip() for l in r.stdout.splitlines() if "Time taken" in l]
                print(f"  default mode list (6 GPU modes + CPU run on 16 threads), {label}: wall {wall:.2f} s, of which CPU bilateral {cpu}", flush=True)
