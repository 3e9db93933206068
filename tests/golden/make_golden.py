"""Regenerates tests/golden/*.npz.  Run from the repo root IN THE BUILD CONTAINER
(needs /root/reference for the _ref part):   python tests/golden/make_golden.py

Two kinds of fixture, kept apart by file name:
  ref_cpu_bilateral_*.npz  inputs + outputs of the REFERENCE's own CPU loop
                           (src/main.cpp:1827-1864 compiled by oracle/Makefile into oracle/_ref).
                           These pin oracle.c::orc_cpu_bilateral and anything checked against it.
  shader_*.npz             inputs + outputs of oracle.c's shader restatements (a1-a5).  The
                           reference offers nothing to pin these ("parity unpinned"): they only
                           freeze the restatement so that an accidental edit of oracle.c, or a
                           compiler/libm change, is detected.
A fixture is data (seeded inputs, expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import synth_hdr, synth_ldr  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    oracle.build()
    if not oracle.have_ref():
        raise SystemExit("oracle/_ref is missing: run in the container that has /root/reference")
    rng = np.random.default_rng(20250205)

    # --- the reference's own loop ------------------------------------------------------------
    for name, (h, w, R), kind in (("a", (40, 48, 10), "hdr"), ("b", (29, 37, 4), "ldr"), ("c", (24, 31, 10), "rand")):
        if kind == "hdr":
            img = synth_hdr(rng, h, w, 2.0)
        elif kind == "ldr":
            img = oracle.unpack_u8(synth_ldr(rng, h, w), flavour=1)
        else:
            img = rng.random((h, w, 4), dtype=np.float32)
        out1 = oracle.ref_cpu_bilateral(img, R, threads=1)
        out8 = oracle.ref_cpu_bilateral(img, R, threads=8)
        assert np.array_equal(out1, out8)
        np.savez_compressed(os.path.join(OUT, f"ref_cpu_bilateral_{name}.npz"), img=img, radius=R, out=out1)

    # --- shader restatements (unpinned; regression only) -------------------------------------
    h, w = 29, 37
    hdr = synth_hdr(rng, h, w, 3.0)
    ldr = synth_ldr(rng, h, w)
    layers = [synth_ldr(rng, h, w) for _ in range(2)]
    nb = (hdr * rng.gamma(16.0, 1 / 16.0, (h, w, 1))).astype(np.float32)
    W0 = np.zeros((h, w, 8), np.float32)
    fix = dict(hdr=hdr, ldr=ldr, layer0=layers[0], layer1=layers[1], nb=nb)
    fix["bil_tex_r4"] = oracle.bilateral_texture(hdr, 4, 2.0, 0.2)
    fix["bil_lin_r4"] = oracle.bilateral_linear(hdr, 4, 2.0, 0.2)
    fix["bil_tex_r8_ldr"] = oracle.bilateral_texture(oracle.unpack_u8(ldr, 0), 8, 2.0, 0.2)
    Wl = oracle.bilateral_layers_accum(hdr, layers[0], W0, 4, 2.0, 0.2)
    Wl = oracle.bilateral_layers_accum(hdr, layers[1], Wl, 4, 2.0, 0.2)
    fix["layers_W"] = Wl
    fix["layers_out"] = oracle.normalize(Wl)
    hs = (hdr * 0.25).astype(np.float32)
    ns = (nb * 0.25).astype(np.float32)
    fix["nlm_in_t"], fix["nlm_in_n"] = hs, ns
    Wn = oracle.nlm_accum(hs, ns, W0, 0.5, (-7, 7), (-3, 3))
    fix["nlm_ref_W"] = Wn
    fix["nlm_bench_W"] = oracle.nlm_accum(hs, ns, W0, 0.5, (-10, 11), (-3, 4))
    fix["nlm_ref_out"] = oracle.normalize(Wn)
    np.savez_compressed(os.path.join(OUT, "shader_restatement.npz"), **fix)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
