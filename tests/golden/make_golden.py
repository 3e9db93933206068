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


def save(name, **arrays):
    """np.savez_compressed, but an existing fixture with identical content is left alone (zip members carry
    timestamps, so rewriting would change the file's bytes for nothing)."""
    path = os.path.join(OUT, name)
    if os.path.exists(path):
        old = np.load(path)
        if sorted(old.files) == sorted(arrays) and all(np.array_equal(old[k], np.asarray(v)) for k, v in arrays.items()):
            return
    np.savez_compressed(path, **arrays)


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
        save(f"ref_cpu_bilateral_{name}.npz", img=img, radius=R, out=out1)

    # --- the reference's loop on BLUE-CONSTANT images ------------------------------------------
    # With blue constant the loop's typo `texColor.b - texColor.b` (src/main.cpp:1850) is also the true
    # blue difference, so the loop IS the shaders' bilateral formula (sigma_s=10, sigma_c=0.2) on the
    # interior -- its output can be held directly against the GPU kernels (a1, a2, a3), no oracle between.
    rb = np.random.default_rng(20251004)
    for name, (h, w, R), kind in (("d", (64, 80, 10), "hdr"), ("e", (64, 80, 4), "ldr"), ("f", (45, 53, 10), "hdr"),
                                  ("g", (45, 53, 4), "ldr")):
        if kind == "hdr":
            img = synth_hdr(rb, h, w, 2.0)
            img[..., 2] = 0.375
            u8 = np.zeros((0,), np.uint8)
        else:
            u8 = synth_ldr(rb, h, w)
            u8[..., 2] = 96                                  # constant blue code, alpha is 255 already
            img = oracle.unpack_u8(u8, flavour=1)            # the CPU path's decode, src/main.cpp:1804-1807
        out1 = oracle.ref_cpu_bilateral(img, R, threads=1)
        assert np.array_equal(out1, oracle.ref_cpu_bilateral(img, R, threads=8))
        save(f"ref_blue_const_{name}.npz", img=img, img_u8=u8, radius=R, out=out1)

    # --- BASELINE configs[0]: 512x512 8-bit PNG, CPU bilateral r=4 ------------------------------
    # Input as the reference decodes a PNG for the CPU path (c * (1/255), :1804-1807), the loop's float output
    # (kept as a SHA-256 of its bits plus every 32nd row -- 4 MB of floats is not a small fixture) and the PNG
    # the reference would write: the truncating pack of :1905-1911 applied to the loop's output.
    import hashlib
    rc = np.random.default_rng(1)                            # seed 1
    u8 = synth_ldr(rc, 512, 512)
    out = oracle.ref_cpu_bilateral(oracle.unpack_u8(u8, flavour=1), 4, threads=8)
    save("ref_cpu_config0_512.npz", img_u8=u8, radius=4, out_u8=oracle.pack_u8(out), out_rows=out[::32],
         out_sha256=np.frombuffer(hashlib.sha256(out.tobytes()).digest(), np.uint8))

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
    save("shader_restatement.npz", **fix)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
