"""A/B of compiler scheduling options for csrc/nlm.hip (development only): builds build/abl/libmi_f<N>.so with
alternative -mllvm flag sets; time them with MID_LIB_PATH=... python tools/ab_nlm.py 0."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {
    0: ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],                                   # shipped
    1: ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    2: ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-misched-prera-direction=topdown"],
    3: ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-misched-prera-direction=bottomup"],
    4: ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-misched-postra"],
    5: ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-set-wave-priority"],
    6: ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-schedule-metric-bias=0"],
    7: ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-misched-cluster=false"],
    8: ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
}
objs = [os.path.join(ROOT, "build", o) for o in
        "capi.cpp.o hostcopy.cpp.o markers.cpp.o recording.cpp.o nlm_small.hip.o pointwise.hip.o bilateral.hip.o nlm_rt.hip.o nlm_rt4.hip.o pipeline.cpp.o sharded.cpp.o codec/png.cpp.o codec/exr.cpp.o codec/piz.cpp.o codec/image_capi.cpp.o".split()]
base = "-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize".split()
d = os.path.join(ROOT, "build", "abl")
os.makedirs(d, exist_ok=True)
src = os.path.join(ROOT, "image_denoising_filter_amd/csrc/nlm.hip")


def build(n):
    o = os.path.join(d, "nlm_f%d.o" % n)
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + base + SETS[n] + ["-I" + os.path.join(ROOT, "include"), "-c", src, "-o", o],
                       capture_output=True, text=True)
    if r.returncode:
        return n, "compile failed: " + r.stderr[-300:]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libmi_f%d.so" % n)] + objs + [o, "-lz", "-ldl"], check=True)
    return n, "ok " + " ".join(SETS[n])


with ThreadPoolExecutor(4) as ex:
    for n, msg in ex.map(build, [int(x) for x in sys.argv[1:]] or sorted(SETS)):
        print(n, msg, flush=True)
