"""Development aid: the CLI at the reference's defaults on a synthetic 1080p animation directory
(4 frames + 4 RenderElements layers per frame), PNG and EXR; prints its console output and wall time."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_denoising_filter_amd as mid
from conftest import synth_hdr, synth_ldr
root = tempfile.mkdtemp()
rng = np.random.default_rng(0)
for ext in ("png", "exr"):
    d = os.path.join(root, ext, "Anim"); os.makedirs(os.path.join(d, "RenderElements"))
    base = synth_hdr(rng, 1080, 1920, 2.0) * 0.25
    for i in range(4):
        f = (np.roll(base, 2 * i, axis=1) * rng.gamma(16.0, 1 / 16.0, (1080, 1920, 1))).astype(np.float32)
        mid.save_image(os.path.join(d, f"Animation01_X_{i:04d}.{ext}"), f if ext == "exr" else (np.clip(f, 0, 1) * 255).astype(np.uint8))
        for name in ("albedo", "normal", "depth", "id"):
            mid.save_image(os.path.join(d, "RenderElements", f"{name}_{i:04d}.png"), synth_ldr(rng, 1080, 1920))
    t0 = time.time()
    r = subprocess.run([os.path.join(os.path.dirname(mid.LIB_PATH), "mi_denoise"), os.path.join(d, f"Animation01_X_0001.{ext}"), "--gpu-only",
                        "--outdir", os.path.join(root, ext)], capture_output=True, text=True)
    print(f"==== {ext}: exit {r.returncode}, wall {time.time() - t0:.2f} s")
    print("\n".join(l for l in r.stdout.splitlines() if "time" in l or "Running" in l))
    print(sorted(f for f in os.listdir(os.path.join(root, ext)) if f.startswith("output")))
    t0 = time.time()
    r = subprocess.run([os.path.join(os.path.dirname(mid.LIB_PATH), "mi_denoise"), os.path.join(d, f"Animation01_X_0000.{ext}"), "--animation",
                        "--temporal-k", "1", "--outdir", os.path.join(root, ext)], capture_output=True, text=True)
    print(f"==== {ext} --animation: exit {r.returncode}, wall {time.time() - t0:.2f} s")
    print("\n".join(l for l in r.stdout.splitlines() if "Mpixel" in l or "time" in l))
