"""Ablation builds of the NLM strip kernel (development only): each variant removes ONE phase of the
per-offset work so that its exposed cost can be timed on the GPU (results are wrong by construction).
Writes build/abl/libmi_abl<N>.so; time them with MID_LIB_PATH=... tools/ab_nlm.py 0.
The patterns address the search-column-innermost loop of rounds 1-2 (`compute()`), so the variants are built with
-DMID_NLM_WALK=0; the kernel text lives in csrc/nlm_strip.hpp, a patched copy of which is written next to a copy of nlm.hip."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image_denoising_filter_amd/csrc")
src = open(os.path.join(CSRC, "nlm_strip.hpp")).read()
tu = open(os.path.join(CSRC, "nlm.hip")).read()
ABL = {
    1: [("for (int k = 0; k < R; ++k) ww[k] = __builtin_amdgcn_exp2f(-dd[k]);    // exp(-d/h^2), nonlocal.comp:55 (d carries log2(e)/h^2)", "for (int k = 0; k < R; ++k) ww[k] = -dd[k];")],
    2: [("for (int k = 0; k < R; ++k) dd[k] = horizontal_box<PLO, PHI>(V[k]);\n            phase(P2{}, P3{});", "for (int k = 0; k < R; ++k) dd[k] = V[k];\n            phase(P2{}, P3{});")],
    3: [("D[m] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));", "D[m] = dx;"),
        ("const float dx = Tr[m] - n[m].x, dy = Tg[m] - n[m].y, dz = Tb[m] - n[m].z;", "const float dx = Tr[m] - n[m].x;")],
    4: [("acc[k].x = fmaf(c.x, wt, acc[k].x); acc[k].y = fmaf(c.y, wt, acc[k].y);   // :56", "acc[k].x = fmaf(c.x + c.y + c.z + c.w, wt, acc[k].x);"),
        ("acc[k].z = fmaf(c.z, wt, acc[k].z); acc[k].w = fmaf(c.w, wt, acc[k].w);", "")],
    5: [("vertical_box<PW, R>(D, V);", "for (int k = 0; k < R; ++k) V[k] = D[k + NL] + D[k];")],
    6: [("load(n, rowp + sx);", "load(n, rowp);")],
}
objs = [os.path.join(ROOT, "build", o) for o in
        "capi.cpp.o pointwise.hip.o bilateral.hip.o nlm_rt.hip.o nlm_rt4.hip.o pipeline.cpp.o sharded.cpp.o codec/png.cpp.o codec/exr.cpp.o codec/piz.cpp.o codec/image_capi.cpp.o".split()]
flags = "-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -DMID_NLM_WALK=0".split()
for n in [int(x) for x in sys.argv[1:]] or sorted(ABL):
    s = src
    for old, new in ABL[n]:
        assert old in s, old
        s = s.replace(old, new)
    d = os.path.join(ROOT, "build", "abl")
    os.makedirs(d, exist_ok=True)
    vd = os.path.join(d, "abl%d" % n)
    os.makedirs(vd, exist_ok=True)
    open(os.path.join(vd, "nlm_strip.hpp"), "w").write(s)         # found first: `#include "nlm_strip.hpp"` looks beside the including file
    p = os.path.join(vd, "nlm.hip")
    open(p, "w").write(tu)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "image_denoising_filter_amd/csrc")]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + inc + ["-c", p, "-o", p + ".o"], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libmi_abl%d.so" % n)] + objs + [p + ".o", "-lz", "-ldl"], check=True)
    print("built", n)
