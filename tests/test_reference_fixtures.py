"""CPU suite: everything that can be held against output of the REFERENCE ITSELF (its CPU bilateral loop,
src/main.cpp:1827-1864, compiled from where it lies into oracle/_ref and run in the build container; the
fixtures under tests/golden/ref_*.npz are its inputs and outputs, made by tests/golden/make_golden.py).

  * a7: oracle.cpu_bilateral == the reference loop bit for bit (new fixtures d-g and the 512x512 config).
  * BASELINE configs[0]: `mi_denoise <512x512.png> --cpu-only --cpu-radius 4` writes exactly the PNG the
    reference's CPU path would (decode c*(1/255), loop, truncating pack -- src/main.cpp:1804-1807,1819-1865,1905-1911).
  * a1/a2 (oracle side): with blue held constant the loop's `texColor.b - texColor.b` typo (:1850) is also the
    true blue difference, so the reference loop computes the shaders' bilateral formula on the interior; the
    oracle's restatements of bialteral.comp / bialteral_linear.comp must agree with the REFERENCE-RUN output there.
    (The GPU kernels are compared with the same fixtures directly in tests/test_gpu_reference_anchor.py.)
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, ROOT, rel_err

BLUE = ["d", "e", "f", "g"]
CLI = os.path.join(ROOT, "image_denoising_filter_amd", "mi_denoise")


def _fix(name):
    g = np.load(os.path.join(GOLDEN, f"ref_blue_const_{name}.npz"))
    return g["img"], g["img_u8"], int(g["radius"]), g["out"]


@pytest.mark.parametrize("name", BLUE)
def test_oracle_cpu_loop_equals_reference_run_output(name):
    img, u8, R, out = _fix(name)
    if u8.size:
        assert np.array_equal(oracle.unpack_u8(u8, flavour=1), img)      # the CPU path's decode
        assert np.all(u8[..., 2] == u8[0, 0, 2])
    assert np.all(img[..., 2] == img[0, 0, 2]), "fixture must be blue-constant"
    for threads in (1, 4):
        assert np.array_equal(oracle.cpu_bilateral(img, R, 10.0, 0.2, True, threads), out)
    # with blue constant the typo is invisible: fixing it changes nothing (except in the last two processed rows,
    # whose windows read the zero pixels past the end of the image: row h, and column w of row h-1)
    h = img.shape[0]
    assert np.array_equal(oracle.cpu_bilateral(img, R, 10.0, 0.2, False, 1)[:h - R - 1], out[:h - R - 1])
    if oracle.have_ref():
        assert np.array_equal(oracle.ref_cpu_bilateral(img, R, threads=2), out)


@pytest.mark.parametrize("name", BLUE)
def test_oracle_shader_restatements_agree_with_reference_run_output_in_the_interior(name):
    """oracle.bilateral_texture / bilateral_linear (restating bialteral.comp:29-73 / bialteral_linear.comp:29-72) vs the
    reference loop's own output: same formula on the interior once blue is constant; float sums vs the loop's
    double-evaluated weights differ by rounding only (tolerance of SURVEY.md 8c: 1e-5)."""
    img, _, R, out = _fix(name)
    h, w = img.shape[:2]
    tex = oracle.bilateral_texture(img, R, 10.0, 0.2)
    lin = oracle.bilateral_linear(img, R, 10.0, 0.2)
    # the loop runs y in [R, h-R], x in [R, w-R] INCLUSIVE (src/main.cpp:1824,1828); column w is read as the next
    # row's first pixel -- exactly the flat-index rule of the linear shader, so the linear restatement matches up
    # to and including column w-R; the texture one where no tap leaves the image.  (Rows h-R-1 and h-R read zero
    # pixels past the end of the image, where the typo and the true blue difference part ways: left out.)
    assert rel_err(lin[R:h - R - 1, R:w - R + 1, :3], out[R:h - R - 1, R:w - R + 1, :3]) < 1e-5
    assert rel_err(tex[R:h - R, R:w - R, :3], out[R:h - R, R:w - R, :3]) < 1e-5
    # alpha: the loop forces 1.0 (:1863); the shaders carry alpha like colour -> 1.0 where every tap has alpha 1
    assert rel_err(tex[R:h - R, R:w - R, 3], out[R:h - R, R:w - R, 3]) < 1e-5


def test_config0_fixture_is_what_the_oracle_computes():
    g = np.load(os.path.join(GOLDEN, "ref_cpu_config0_512.npz"))
    u8, R = g["img_u8"], int(g["radius"])
    assert u8.shape == (512, 512, 4) and R == 4
    out = oracle.cpu_bilateral(oracle.unpack_u8(u8, flavour=1), R, 10.0, 0.2, True, 8)
    assert hashlib.sha256(out.tobytes()).digest() == g["out_sha256"].tobytes(), "float output differs from the reference run"
    assert np.array_equal(out[::32], g["out_rows"])
    assert np.array_equal(oracle.pack_u8(out), g["out_u8"])


def test_config0_cli_writes_the_references_png(tmp_path):
    """BASELINE configs[0] end to end through the drop-in CLI, against reference-run output (no oracle between)."""
    import image_denoising_filter_amd as mid
    g = np.load(os.path.join(GOLDEN, "ref_cpu_config0_512.npz"))
    src = tmp_path / "frame_0000.png"
    mid.save_image(src, g["img_u8"])
    assert np.array_equal(mid.load_image(src), g["img_u8"])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    for threads in ("1", "8"):
        r = subprocess.run([CLI, str(src), "--cpu-only", "--cpu-radius", "4", "--cpu-threads", threads], cwd=tmp_path,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        got = mid.load_image(tmp_path / "output-cpu.png")
        assert got.shape == (512, 512, 4) and np.array_equal(got, g["out_u8"]), f"{threads} thread(s)"


@pytest.mark.parametrize("patch", [(-3, 4), (-3, 3)])
def test_oracle_nlm_on_a_linear_ramp_is_the_pinned_bilateral_formula(patch):
    """The same identity as tests/test_gpu_reference_anchor.py::test_a4_..., on the checker: on a linear ramp
    oracle.nlm_accum (restating nonlocal.comp) must equal oracle.bilateral_texture (pinned to reference-run output above)
    with sigma_c = h / sqrt(2 P^2) and no spatial term, up to the 0.001 norm bias."""
    h, w, R, hp = 40, 52, 5, 0.5
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.2 + 0.004 * xx + 0.001 * yy, 0.9 - 0.002 * xx + 0.003 * yy, 0.5 + 0.0015 * xx - 0.0025 * yy,
                    np.ones_like(xx)], -1).astype(np.float32)
    P2 = (patch[1] - patch[0]) ** 2
    Wn = oracle.nlm_accum(img, img, np.zeros((h, w, 8), np.float32), hp, (-R, R + 1), patch, threads=4)
    bil = oracle.bilateral_texture(img, R, 1e6, hp / np.sqrt(2.0 * P2))
    m = R + max(-patch[0], patch[1])
    sw = Wn[m:-m, m:-m, 4].astype(np.float64) - 0.001
    assert rel_err(Wn[m:-m, m:-m, :4].astype(np.float64) / sw[..., None], bil[m:-m, m:-m]) < 2e-5
