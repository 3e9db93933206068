"""CPU suite: bench.py's read-back hardware counters are tied to the code of the kernel that is loaded.

`roofline.traffic` / `valu_util` cannot be measured inside an un-profiled run; bench.py reads them from the newest
profiles/r*_traffic.json / r*_utilisation.json.  Since round 5 those files carry the sha256 of the profiled kernel's machine
code (function bytes + kernel descriptor, image_denoising_filter_amd/_codeobj.py), bench.py recomputes it from the library
it loads, and a profile without a fingerprint or with another one yields null + the reason -- never a stale number."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "image_denoising_filter_amd", "libmi_denoise.so")


def _codeobj():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_codeobj_under_test", os.path.join(ROOT, "image_denoising_filter_amd", "_codeobj.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_fingerprint_identifies_one_kernel_of_the_built_library():
    co = _codeobj()
    fp = co.fingerprint(LIB, "nlm")
    assert len(fp["kernel_code_sha256"]) == 64 and fp["kernel_code_bytes"] > 10_000          # ~30 KB of unrolled offsets
    assert co.fingerprint(LIB, "nlm") == fp                                                  # a function of the file alone
    # a different instantiation (the temporal one: MULTI = true) has different code, hence a different fingerprint
    other = co.BENCH_KERNELS["nlm"].replace("Lb1ELb0ELb0E", "Lb1ELb1ELb0E")
    assert co.kernel_sha256(LIB, other)[0] != fp["kernel_code_sha256"]
    with pytest.raises(ValueError):
        co.kernel_sha256(LIB, "_ZN3mid9no_such_kernelEv")
    objs = co.gfx950_code_objects(open(LIB, "rb").read())
    assert len(objs) >= 4 and all(o[:4] == b"\x7fELF" for o in objs)                         # one code object per .hip translation unit


def test_counters_are_read_back_only_for_the_loaded_kernels_code(tmp_path, monkeypatch):
    import bench
    co = _codeobj()
    digest = co.fingerprint(LIB, "nlm")["kernel_code_sha256"]
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    F = 31
    alg = F * bench.NPIX * bench.NLM_BYTES_PER_PX
    base = {"traffic_bytes_per_launch": 2.2e9, "algorithmic_bytes_per_launch": alg, "date": "2026-10-05"}
    util = {"date": "2026-10-05", "kernels": {"nlm_bench": {"valu_issue_util": 0.84, "lds_util": 0.05, "valu_active_share_of_wave_cycles": 0.8,
                                                             "issue_stall_share_of_wave_cycles": 0.1, "lds_bank_conflict_cycles": 0}}}
    fp = bench.loaded_kernel_fingerprint(LIB, "nlm")
    assert fp == (digest, None)

    (prof / "r09_traffic.json").write_text(json.dumps(base))                                  # a pre-round-5 profile: no fingerprint
    (prof / "r09_utilisation.json").write_text(json.dumps(util))
    v, why = bench.load_traffic(F, fp)
    assert v is None and "no kernel fingerprint" in why
    u = bench.load_utilisation("nlm", fp)
    assert u["valu_util"] is None and "no kernel fingerprint" in u["utilisation_source"]

    (prof / "r09_traffic.json").write_text(json.dumps(dict(base, kernel_code_sha256="0" * 64)))      # profiled on other code
    (prof / "r09_utilisation.json").write_text(json.dumps(dict(util, bench_kernel_code_sha256="0" * 64)))
    v, why = bench.load_traffic(F, fp)
    assert v is None and "kernel changed" in why and digest[:12] in why
    assert bench.load_utilisation("nlm", fp)["valu_util"] is None

    (prof / "r09_traffic.json").write_text(json.dumps(dict(base, kernel_code_sha256=digest)))        # the loaded kernel's own profile
    (prof / "r09_utilisation.json").write_text(json.dumps(dict(util, bench_kernel_code_sha256=digest)))
    v, why = bench.load_traffic(F, fp)
    assert v == 2200000000 and "matches the loaded library" in why
    assert bench.load_utilisation("nlm", fp)["valu_util"] == 0.84
    v, why = bench.load_traffic(16, fp)                                                       # another launch shape
    assert v is None and "different launch shape" in why
    # a library whose fingerprint cannot be computed: nothing is read back either
    v, why = bench.load_traffic(F, bench.loaded_kernel_fingerprint(str(tmp_path / "missing.so"), "nlm"))
    assert v is None and "fingerprint of the loaded library failed" in why


def test_the_committed_profiles_either_match_the_built_kernel_or_say_why_not():
    """Whatever is committed under profiles/ right now, bench.py's answer is a number with a matching fingerprint or null with a
    reason -- and the NEWEST traffic profile is expected to match the tree's kernel (refresh with tools/run_profiles.sh)."""
    import bench
    fp = bench.loaded_kernel_fingerprint(LIB, "nlm")
    v, why = bench.load_traffic(31, fp)
    assert (v is None) != ("matches the loaded library" in why), (v, why)


def test_configs0_image_of_the_cpu_legs_is_the_stated_one():
    import bench
    img = bench.synth_c1()
    assert img.shape == (512, 512, 4) and img.dtype.name == "uint8" and (img[..., 3] == 255).all()
    assert (bench.synth_c1() == img).all() and img[..., :3].std() > 20                        # seeded; gradient + discs + noise


def test_the_sharded_paths_report_is_assembled_for_worlds_this_pool_cannot_run():
    """bench.py's `scaling_strong` / `also.temporal_nlm_k2_native` for N > 1 come from rows gathered over the ranks; the assembly is a
    pure function so that its N = 2 and N = 8 forms are exercised here (the driver's multi-GPU run is the first to execute them on
    hardware): worst-rank halo_hidden_frac, bytes on the wire, per-rank lists of the right length, JSON-serialisable."""
    import bench
    for world in (2, 8):
        frames = [64 // world] * world
        rows = []
        for r in range(world):
            edge = r in (0, world - 1)
            recv = (2 if edge else 4) * bench.NPIX * 16
            rows.append([0.020 + 0.001 * r, float(recv), float(recv), 0.45, 1.0, 0.02, 0.47, 9.0 + r, 19.0, 1.0 if r else 0.8, float(world), float(r)])
        nat, strong = bench.native_temporal_report(rows, 64, world, False, "XIWBWB", (-1, 1, -1), 22606, frames)
        json.dumps({"a": nat, "b": strong})
        assert strong["n_gpus"] == world and strong["scaling"] == "strong" and strong["frames_per_rank"] == frames
        assert strong["value"] == round(64 * bench.NPIX / 1e6 / (0.020 + 0.001 * (world - 1)), 1)            # max over ranks sets the time
        assert strong["halo_hidden_frac"] == 0.8 and len(strong["halo_hidden_frac_per_rank"]) == world       # the worst rank's
        assert strong["bytes_on_the_wire"] == sum(int(r[2]) for r in rows) == (2 * (world - 1)) * 2 * bench.NPIX * 16
        assert strong["rccl_comm_count"] == [world] * world and strong["hardware_status"] == "measured in this run"
        assert all(strong["bit_identical_to_single_launch_per_rank"]) and nat["rccl"]["user_rank_per_rank"] == list(range(world))
        assert nat["timeline_ms_per_rank"][1] == {"exchange_start": 0.02, "exchange_end": 0.47, "interior_end": 10.0, "end": 19.0}
    nat, strong = bench.native_temporal_report([[0.14, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 140.0, 140.0, -1.0, 1.0, 0.0]], 64, 1, False, "I", (-1, 1, -1), 22606, [64])
    assert strong["halo_hidden_frac"] is None and "unmeasured on hardware" in strong["hardware_status"] and strong["bytes_on_the_wire"] == 0
