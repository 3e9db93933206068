#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the denoise hot path on MI355X.

Metric (BASELINE.json): Mpixel/s of non-local means, 21x21 search / 7x7 patch, on 1920x1080
RGBA32F frames resident in HBM.  A "step" is one pass of the hot path over one batch of
`--frames` synthetic frames per GPU (one fused launch: accumulate + normalize per frame;
default 31 = a whole number of rounds of workgroups on the chip, see --frames below).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); single-frame NLM shards by
frame with no data-path collective, so every rank filters its own batch ("weak" scaling) and
value = all ranks' pixels / max-over-ranks time.  After the timed region (outside it) the run
also measures, and reports under "also": bilateral r=8 in both layouts (BASELINE configs[1]), the
temporal +-2 NLM with its RCCL halo exchange (configs[4]) and the PCIe-inclusive pipeline rate.

Prints ONE JSON line (rank 0).  The oracle is used only for the cpu_baseline leg.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# dmabuf IPC between the per-GPU processes: this pool's host driver supports nothing else (the build brief's environment
# notes: "without it RCCL / CUDA-tensor sharing across processes fails with hipIpcGetMemHandle: invalid argument"; the
# variable is already exported on the GPU boxes).  Set here, before anything can initialise the HSA runtime, so that the
# self-launcher, `torch.distributed.run ... bench.py` and a plain 1-GPU run all see the same setting.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
NPIX = W * H
SEARCH, PATCH = (-10, 11), (-3, 4)        # 21x21 / 7x7, half-open
HPARAM = 0.5
# algorithmic work per pixel: minimum-work NLM = 32 flop per (pixel, offset)
NLM_FLOP_PER_PX = 32 * 21 * 21            # 14,112
NLM_BYTES_PER_PX = 32                     # 16 B read + 16 B written (fused accumulate+normalize)
BIL_FLOP_PER_PX = 20 * 17 * 17            # 5,780 at r=8
BIL_BYTES_PER_PX = 32
PEAK_FP32_TFLOPS = 157.3                  # MI355X_MICROARCH.md: fp32 vector = dense f32 MFMA peak
PEAK_HBM_GBS = 8000.0
SEQ_FRAMES = 64                           # BASELINE configs[4]: one 64-frame animation, temporal +-2
WATCHDOG_S = 300.0                        # the side measurements after the timed region may take this long before the line is forced out


def synth_frames(n, seed, device, shift=0):
    """Seeded 'path-tracer-like' HDR frames: piecewise-smooth radiance with
    highlights times per-pixel Gamma(4) noise, alpha = 1.  Generated on the GPU with torch."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    base = torch.stack([0.6 + 0.4 * torch.sin(xx * 0.011 + 1.0), 0.5 + 0.5 * torch.cos(yy * 0.007),
                        0.4 + 0.3 * torch.sin((xx + yy) * 0.005)], -1)
    hl = torch.exp(-(((xx - 0.6 * W) ** 2 + (yy - 0.3 * H) ** 2) / (0.02 * W * H)))[..., None] * 6.0
    out = []
    for i in range(n):
        # Gamma(4, 1/4) as the mean of 4 exponentials; pan 2 px per frame
        u = torch.rand((4, H, W, 1), generator=g, device=device).clamp_min(1e-7)
        noise = (-torch.log(u)).mean(0)
        rad = torch.roll(base + hl, shifts=2 * i + shift, dims=1) * noise
        out.append(torch.cat([rad, torch.ones((H, W, 1), device=device)], -1).contiguous())
    return out


def loaded_kernel_fingerprint(lib_path, workload):
    """sha256 of the timed kernel's machine code in the library this process has LOADED (function bytes + kernel descriptor,
    image_denoising_filter_amd/_codeobj.py), or (None, reason).  The counter figures below are only read back from
    profiles/ when the profile was taken on a kernel with the same fingerprint."""
    try:
        from image_denoising_filter_amd import _codeobj
        if workload not in _codeobj.BENCH_KERNELS:
            return None, f"no fingerprint defined for workload {workload!r}"
        return _codeobj.fingerprint(lib_path, workload)["kernel_code_sha256"], None
    except Exception as e:  # noqa: BLE001
        return None, f"fingerprint of the loaded library failed: {e}"


def _same_code(profile, key, name, fp):
    """None when the profile `name` was taken on the kernel this run has loaded, else the reason it may not be read back."""
    digest, why = fp
    if digest is None:
        return why
    if not profile.get(key):
        return f"{name} carries no kernel fingerprint (taken before round 5): refusing to read counters of an unidentified kernel back"
    if profile[key] != digest:
        return (f"{name} was measured on kernel code {profile[key][:12]}..., the loaded library holds {digest[:12]}...: "
                "the kernel changed after the profile -- refresh with tools/run_profiles.sh")
    return None


def load_traffic(frames_per_launch, fp, workload="nlm"):
    """(HBM bytes per launch, where that figure comes from).  PMC counters cannot be read from inside an
    un-profiled run, so the figure is the one measured by the committed rocprofv3 `--pmc` passes of this same
    command (profiles/*_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); `traffic_source` in the JSON
    line says so, with the file and its date.  (None, reason) when no profile matches the launch shape or the
    profile was taken on different kernel code than the library now loaded holds (`fp`)."""
    import glob
    pattern = "r*_traffic.json" if workload == "nlm" else f"r*_traffic_{workload}.json"
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not cands:
        return None, f"no profiles/{pattern}"
    try:
        t = json.load(open(cands[-1]))
        stale = _same_code(t, "kernel_code_sha256", os.path.basename(cands[-1]), fp)
        if stale:
            return None, stale
        if t.get("algorithmic_bytes_per_launch") != frames_per_launch * NPIX * NLM_BYTES_PER_PX:   # (32 B/px for both workloads)
            return None, f"{os.path.basename(cands[-1])} was measured for a different launch shape"
        when = t.get("date") or time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(cands[-1])))
        return round(t["traffic_bytes_per_launch"]), (f"profiles/{os.path.basename(cands[-1])} ({when}): rocprofv3 --pmc FETCH_SIZE / "
                                                     "WRITE_SIZE passes of this command, read back -- not measured in this run; "
                                                     f"kernel code sha256 {t['kernel_code_sha256'][:12]}... matches the loaded library")
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable profile: {e}"


def load_utilisation(workload, fp):
    """Counter-derived utilisation of the dominant kernel (VALU issue, LDS, stall share), read back from the committed
    profiles/r*_utilisation.json exactly like `traffic` -- PMC counters cannot be read inside an un-profiled run.
    tools/summarize_profiles.py writes that file from the rocprofv3 --pmc passes of this command and states every
    unit; `utilisation_source` names the file and its date."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_utilisation.json")))
    if not cands:
        return {"valu_util": None, "lds_util": None, "utilisation_source": "no profiles/r*_utilisation.json"}
    try:
        u = json.load(open(cands[-1]))
        by = dict(u.get("bench_kernel_code_sha256_by_workload") or {})
        by.setdefault("nlm", u.get("bench_kernel_code_sha256"))
        stale = _same_code({"h": by.get(workload)}, "h", os.path.basename(cands[-1]), fp)
        if stale:
            return {"valu_util": None, "lds_util": None, "utilisation_source": stale}
        k = u["kernels"]["nlm_bench" if workload == "nlm" else "bilateral_r8_linear"]
        return {"valu_util": k["valu_issue_util"], "valu_util_at_occupancy_prices": k.get("valu_issue_util_at_occupancy"),
                "cycles_per_wave_offset": {q: k["cycles_per_wave_offset"][q] for q in ("measured", "floor", "at_occupancy_prices")} if k.get("cycles_per_wave_offset") else None,
                "lds_util": k["lds_util"],
                "valu_active_share_of_wave_cycles": k["valu_active_share_of_wave_cycles"],
                "issue_stall_share_of_wave_cycles": k["issue_stall_share_of_wave_cycles"],
                "lds_bank_conflict_cycles": k["lds_bank_conflict_cycles"],
                "utilisation_source": f"profiles/{os.path.basename(cands[-1])} ({u.get('date')}): rocprofv3 --pmc passes of this "
                                      "command, read back -- not measured in this run; definitions and units in that file"}
    except Exception as e:  # noqa: BLE001
        return {"valu_util": None, "lds_util": None, "utilisation_source": f"unreadable profile: {e}"}


class Timers:
    """hipEvent pairs recorded on the launch stream (mid_timer_*), read after the final sync."""

    def __init__(self, mid, ctx, n):
        self.lib, self.ctx = mid.lib, ctx
        self.t = []
        for _ in range(n):
            p = ctypes.c_void_p()
            assert self.lib.mid_timer_create(ctx.handle, ctypes.byref(p)) == 0
            self.t.append(p)

    def tick(self, i, stream):
        self.lib.mid_timer_tick(self.t[i], stream)

    def tock(self, i, stream):
        self.lib.mid_timer_tock(self.t[i], stream)

    def ms(self):
        out = []
        for p in self.t:
            v = ctypes.c_float()
            assert self.lib.mid_timer_ms(p, ctypes.byref(v)) == 0
            out.append(v.value)
        return out

    def close(self):
        for p in self.t:
            self.lib.mid_timer_destroy(p)
        self.t = []


def synth_c1(seed=1):
    """SURVEY.md 8d C1 = BASELINE configs[0]: 512x512 RGBA8, smooth gradient + 3 hard-edged discs + Gaussian noise sigma 10/255."""
    rng = np.random.default_rng(seed)
    n = 512
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float32)
    img = np.stack([xx / (n - 1), yy / (n - 1), 0.5 + 0.25 * (xx + yy) / (n - 1)], -1)
    for cx, cy, r, col in ((0.3, 0.35, 0.16, (0.9, 0.2, 0.1)), (0.7, 0.6, 0.2, (0.1, 0.8, 0.3)), (0.45, 0.8, 0.1, (0.2, 0.3, 0.9))):
        img[(xx - cx * n) ** 2 + (yy - cy * n) ** 2 < (r * n) ** 2] = col
    img = np.clip(img + rng.normal(0, 10 / 255, img.shape), 0, 1)
    return (np.concatenate([img, np.ones((n, n, 1))], -1) * 255).astype(np.uint8)


def _cpu_reference_bilateral(oracle):
    """The reference's own CPU bilateral loop (oracle/_ref, src/main.cpp:1827-1864, sigma 10/0.2 are literals of the slice,
    OpenMP over x) on this host's cores -- the only CPU code path the reference has.  SURVEY.md 8d's legs, each a bounded
    sample (the whole table stays inside ~20 s):
      C1  = BASELINE configs[0]: the 512x512 RGBA8 image decoded as RunOnCPU decodes it (c * (1/255), src/main.cpp:1804-1807), r = 4;
      C2  = a strip of the 1920x1080 RGBA32F frame at r = 8 (like for like with the GPU bilateral of configs[1]) and at r = 10
            (the reference's default windowSize, src/main.cpp:1819);
      threads in {1, 8 (the reference's two choices, src/main.cpp:1979,1984), all cores of this box's share}.
    The headline fields (value, cores, sample) stay what they were: r = 10, 8 threads, on a 1920x360 strip.
    `oracle` is the checker module, handed in by cpu_baseline() -- the one place that imports it."""
    kind = "reference" if oracle.have_ref() else "port"

    def run(img, radius, threads):
        if kind == "reference":
            return oracle.ref_cpu_bilateral(img, radius, threads)
        return oracle.cpu_bilateral(img, radius, 10.0, 0.2, True, threads)

    nproc = min(os.cpu_count() or 1, 16)              # 16 = the CPU share of a one-GPU box
    thread_set = sorted({1, min(8, nproc), nproc})
    rng = np.random.default_rng(1)
    frame = (rng.random((360, W, 4), dtype=np.float32) * 4).astype(np.float32)
    c1 = oracle.unpack_u8(synth_c1(), flavour=1)      # CPU decode flavour: (float)c * (1.0f / 255.0f)
    run(frame[:40], 4, nproc)                         # OpenMP team start-up, page faults
    legs = []

    def leg(name, img, radius, threads, budget_s, min_runs=1, max_runs=3):
        ts = []
        t_all = time.perf_counter()
        while len(ts) < min_runs or (len(ts) < max_runs and time.perf_counter() - t_all < budget_s):
            t0 = time.perf_counter()
            run(img, radius, threads)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        h_, w_ = img.shape[:2]
        legs.append({"input": name, "radius": radius, "threads": threads, "Mpixel/s": round(h_ * w_ / 1e6 / med, 4),
                     "sample_pixels": h_ * w_, "runs": len(ts), "seconds_per_run": round(med, 4)})
        return legs[-1]

    # rows per leg sized so that one run is ~0.3-1 s at the rates this loop reaches (0.2 Mpx/s per thread at r=10; cost ~ taps)
    for th in thread_set:
        leg("C1 512x512 RGBA8 (configs[0]), whole image", c1, 4, th, 1.5)
    for radius, taps in ((8, 289), (10, 441)):
        for th in thread_set:
            rows = 360 if th == min(8, nproc) and radius == 10 else max(2 * radius + 8, min(360, int(0.2 * th * 441 / taps * 0.6e6 / W) // 4 * 4))
            leg(f"C2 1920x{rows} RGBA32F strip of the 1080p frame", frame[:rows], radius, th, 2.5)
    # once, for context: the same loop with the reference's OWN compile flags -- "-fopenmp -g", no optimisation level
    # (CMakeLists.txt:31) -- at its default radius and thread count, on a thin strip
    as_shipped = None
    if kind == "reference" and oracle.have_ref_as_shipped():
        strip = frame[:40]
        t0 = time.perf_counter()
        oracle.ref_cpu_bilateral(strip, 10, min(8, nproc), as_shipped=True)
        dt = time.perf_counter() - t0
        as_shipped = {"flags": "-O0 -g -fopenmp (the reference's CMakeLists.txt:31)", "radius": 10, "threads": min(8, nproc),
                      "Mpixel/s": round(40 * W / 1e6 / dt, 4), "sample_pixels": 40 * W, "seconds_per_run": round(dt, 4)}
    head = [l for l in legs if l["radius"] == 10 and l["threads"] == min(8, nproc)][0]
    return {"value": head["Mpixel/s"], "unit": "Mpixel/s", "cores": head["threads"], "kind": kind,
            "legs": legs, "as_shipped_build": as_shipped,
            "sample": f"reference CPU bilateral loop (r=10, sigma_s=10, sigma_c=0.2, {head['threads']} OpenMP threads, -O2) "
                      f"on a 1920x360 RGBA32F strip, median of {head['runs']} runs; `legs` = SURVEY 8d's table "
                      "(C1 r=4, C2 r=8 and r=10, threads 1 / 8 / all)"}


def cpu_baseline(workload="nlm"):
    """CPU figure for the SAME workload as `value`, on a bounded sample, on this host's cores.
    nlm: the oracle's restatement of nonlocal.comp (kind "port": the reference has no CPU NLM), one invocation per
    pixel exactly as the shader, rows spread over OpenMP threads.  bilateral: the reference's own CPU loop."""
    import oracle
    if workload != "nlm":
        return _cpu_reference_bilateral(oracle)
    threads = min(os.cpu_count() or 1, 16)          # 16 = the CPU share of a one-GPU box
    rng = np.random.default_rng(2)
    done, spent, rows = 0, 0.0, 128
    img = (rng.random((rows, W, 4), dtype=np.float32) * 4).astype(np.float32)
    Wz = np.zeros((rows, W, 8), np.float32)
    while spent < 12.0 and done < 64:               # 1920x128 strips (search halo clipped at the strip edge), ~12 s in all
        t0 = time.perf_counter()
        acc = oracle.nlm_accum(img, img, Wz, HPARAM, SEARCH, PATCH, threads=threads)
        oracle.normalize(acc)
        spent += time.perf_counter() - t0
        done += 1
    out = {"value": round(done * rows * W / 1e6 / spent, 5), "unit": "Mpixel/s", "cores": threads, "kind": "port",
           "sample": f"oracle restatement of shaders/nonlocal.comp + normalize.comp (21x21 search, 7x7 patch, h={HPARAM}, "
                     f"scalar C -O2, {threads} OpenMP threads over rows) on {done} strip(s) of 1920x{rows} RGBA32F = "
                     f"{done * rows * W / 1e6:.2f} Mpixel in {spent:.1f} s; the reference itself has no CPU NLM"}
    try:
        out["reference_cpu_path"] = _cpu_reference_bilateral(oracle)
    except Exception as e:  # noqa: BLE001 - the secondary figure must not cost the primary one
        out["reference_cpu_path"] = {"value": None, "sample": f"failed: {e}"}
    # The reference's OWN CPU path, where cpu_baseline is read: value/kind above are the port of the headline workload (the
    # reference has no CPU NLM); `reference` is the loop the reference does ship (bilateral, src/main.cpp:1819-1865), compiled
    # from its sources -- another filter than `value`'s, so it sits beside the port, not in its place.
    ref = out["reference_cpu_path"]
    out["reference"] = {kk: ref.get(kk) for kk in ("value", "unit", "cores", "kind", "sample")}
    out["reference"]["workload"] = "bilateral r=10 (RunOnCPU's own window), not the NLM workload of `value`; all legs under reference_cpu_path"
    return out


def native_temporal_report(rows, n_seq, world, rehearse, order, prio, rc_ver, frames_per_rank):
    """The two objects the C++ RCCL path contributes to the line, from the per-rank rows gathered after the timed repetitions
    (a pure function: tests/test_bench_counters.py feeds it rows of 2 and 8 ranks, a world this pool cannot run).  Row layout:
    [seconds per sequence, halo bytes received, halo bytes sent, exchange ms, bit-identical (1/0), exchange start ms, exchange end ms,
    interior end ms, call end ms, halo_hidden_frac or -1, ncclCommCount, ncclCommUserRank]."""
    te = max(r[0] for r in rows)
    nat = {"Mpixel/s_out": round(n_seq * NPIX / 1e6 / te, 1), "frames": n_seq,
           "ms_per_sequence": round(te * 1e3, 3),
           "halo_bytes_recv_per_rank": [int(r[1]) for r in rows],
           "halo_bytes_sent_per_rank": [int(r[2]) for r in rows],
           "exchange_ms_per_rank": [round(r[3], 4) for r in rows],
           "bit_identical_to_single_launch_per_rank": [bool(r[4]) for r in rows],
           # device timeline of one call, ms from the call's first event on the launch stream, per rank
           "timeline_ms_per_rank": [{"exchange_start": round(r[5], 4), "exchange_end": round(r[6], 4),
                                     "interior_end": round(r[7], 4), "end": round(r[8], 4)} for r in rows],
           "halo_hidden_frac_per_rank": [None if r[9] < 0 else round(r[9], 4) for r in rows],
           "issue_order_rank0": order, "exchange_stream_priority": {"priority": prio[0], "least": prio[1], "greatest": prio[2]},
           "rccl": {"comm_count_per_rank": [int(r[10]) for r in rows], "user_rank_per_rank": [int(r[11]) for r in rows], "version": rc_ver},
           "path": "C++: mid_comm_create (ncclCommInitRank via dlopen) + mid_nlm_temporal_sharded "
                   "(ncclSend/ncclRecv in one group on the highest-priority exchange stream, interior launches meanwhile, "
                   "boundary launches on their own stream)"}
    # The sharded path at the top level of the line (the headline above is replicas: weak scaling of independent batches):
    # BASELINE configs[4] -- ONE 64-frame 1080p sequence, temporal +-2 -- split over the ranks; the same job at every N.
    hid = [x for x in nat["halo_hidden_frac_per_rank"] if x is not None]
    strong = {
        "workload": "nlm_temporal_k2_64_frames_1080p_hdr (BASELINE configs[4]): one sequence, contiguous frame blocks per rank, halo over RCCL",
        "metric": "output Mpixel/s (64 x 1920 x 1080 / max-over-ranks seconds per sequence)",
        "value": nat["Mpixel/s_out"], "unit": "Mpixel/s", "n_gpus": world, "scaling": "strong", "ms_per_sequence": nat["ms_per_sequence"],
        "frames_per_rank": list(frames_per_rank),
        "bit_identical_to_single_launch_per_rank": nat["bit_identical_to_single_launch_per_rank"],
        "bytes_on_the_wire": int(sum(nat["halo_bytes_sent_per_rank"])),
        "rccl_comm_count": nat["rccl"]["comm_count_per_rank"], "rccl_version": rc_ver,
        "exchange_ms_per_rank": nat["exchange_ms_per_rank"],
        "halo_hidden_frac": (round(min(hid), 4) if hid else None),
        "halo_hidden_frac_per_rank": nat["halo_hidden_frac_per_rank"],
        "halo_hidden_frac_def": "share of a rank's exchange (first receive posted .. last transfer complete, device timeline) that ran while "
                                "its interior launches were still executing: (min(exchange_end, interior_end) - exchange_start) / "
                                "(exchange_end - exchange_start); the value is the worst rank's; null with one rank (nothing is exchanged)",
        "hardware_status": ("measured in this run" if world > 1 and not rehearse else
                            "one rank: no exchange took place -- the N >= 2 figures are unmeasured on hardware until a multi-GPU run"),
    }
    return nat, strong


class _DryContext:
    """Stands in for the C-ABI context in --dry-run: records what would be launched, computes nothing."""
    def __init__(self):
        self.launches = []

    def nlm_temporal_dev(self, frame_ptrs, out_ptrs, w, h, hparam, search, patch, k, first, count, fmt, stream=None):
        assert 0 <= first and first + count <= len(frame_ptrs) and len(out_ptrs) == count
        self.launches.append((len(frame_ptrs), k, first, count))


def dry_run(args, json_out):
    """The N-rank control flow of main() on CPU tensors over gloo: process group, barriers, the timed loop,
    max-over-ranks, the temporal step with its overlapped halo exchange, one JSON line from rank 0."""
    from image_denoising_filter_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        world = dist.get_world_size()
    h, w, F, k = 12, 20, args.frames, 2
    g = torch.Generator().manual_seed(100 + rank)
    frames = [torch.rand((h, w, 4), generator=g) for _ in range(F)]
    outs = [torch.empty((h, w, 4)) for _ in range(F)]
    ctx = _DryContext()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.nlm_temporal_dev([f.data_ptr() for f in frames], [o.data_ptr() for o in outs], w, h, HPARAM, SEARCH, PATCH, 0, 0, F, 0)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_seq = world * F
    start, count = sharding.partition(n_seq, world)[rank]
    covered = []

    def launch(fr, first, cnt, off):
        ctx.nlm_temporal_dev([f.data_ptr() for f in fr], [o.data_ptr() for o in outs[off:off + cnt]], w, h, HPARAM, SEARCH, PATCH, k, first, cnt, 0)
        covered.extend(range(off, off + cnt))
    have = sharding.temporal_block_overlapped(launch, frames, n_seq, k)
    assert sorted(covered) == list(range(count)), covered
    lo, hi = max(0, start - k), min(n_seq - 1, start + count - 1 + k)
    assert sorted(have) == list(range(lo, hi + 1))
    if world > 1:
        # halo frames must be the neighbours' data: every rank seeds its frames with 100 + its rank
        for f, tns in have.items():
            owner = f // F
            ref = torch.rand((h, w, 4), generator=torch.Generator().manual_seed(100 + owner)) if f % F == 0 else None
            if ref is not None:
                assert torch.equal(tns, ref), f"frame {f} is not rank {owner}'s data"
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no kernels)", "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "value": None, "elapsed_s": round(elapsed, 6), "launches_rank0": len(ctx.launches)}), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints a version banner on stdout when a
    communicator is created through the C API (seen on the GPU box: 'RCCL version : ...', 'Librccl path : ...') -- so file
    descriptor 1 is pointed at stderr for the whole run and the JSON line is written to a private duplicate of the
    original stdout.  Returns that file object."""
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` started plainly (no torchrun, WORLD_SIZE unset): this process becomes a pure
    launcher.  It never touches the GPU (no torch.cuda call, no HIP call, no exec): it starts N fresh children
    of this same script -- one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment, rendezvous on
    127.0.0.1 -- relays rank 0's single JSON line on stdout (the other ranks' stdout goes to stderr) and
    returns non-zero if any rank fails; when one rank dies the others are terminated (by PID) so that a
    collective waiting for the dead rank cannot hang the run."""
    import subprocess
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    # (HSA_ENABLE_IPC_MODE_LEGACY is set at the top of this module for every way of starting it; the children inherit it)
    env.setdefault("OMP_NUM_THREADS", "1")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py launcher: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    json_lines = [l for l in out0 if l.startswith("{")]
    for l in out0:
        (sys.stdout if l.startswith("{") else sys.stderr).write(l)
    sys.stdout.flush()
    if rc == 0 and len(json_lines) != 1:
        print(f"bench.py launcher: expected one JSON line from rank 0, got {len(json_lines)}", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser(
        description="Headline benchmark.  N > 1: either start it under torch.distributed.run (one rank per GPU, as the "
                    "driver does) or plainly as `python bench.py --gpus N` -- it then launches its own N ranks.")
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node to use, one process each; without RANK/WORLD_SIZE in the environment "
                         "bench.py spawns the N ranks itself")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # 31: a 1080p frame is 1156 workgroups and an MI355X holds 512 at a time (two 76 KB tiles on each of 256 CUs), so a launch over
    # 31 frames is 69.99 rounds of workgroups -- no last, mostly empty round (16 frames: 36.1 rounds, 1.7 % slower per frame;
    # profiles/r03_frames_per_launch_sweep.txt).  The side measurements below keep their 16-frame shapes.
    ap.add_argument("--frames", type=int, default=31, help="frames per GPU per step (one fused launch)")
    ap.add_argument("--workload", choices=["nlm", "bilateral"], default="nlm",
                    help="what the timed region measures: nlm = BASELINE configs[2] (the north_star target, default); "
                         "bilateral = configs[1], r=8, linear-buffer layout, one launch per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank control flow (gloo, no kernels, tiny frames); "
                         "marks its JSON dry_run=true -- never a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher above us: become one, BEFORE anything touches the GPU (see launch_ranks)
        sys.exit(launch_ranks(args.gpus))
    json_out = claim_stdout()
    if args.dry_run:
        return dry_run(args, json_out)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a launcher that disagrees with --gpus is a set-up error, not something to paper over with a 1-GPU number
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # MID_BENCH_REHEARSE=1: a rehearsal of the N-rank flow on a box with FEWER GPUs than ranks -- ranks share devices,
    # the process group is gloo (RCCL refuses two ranks on one device), halo frames are staged through host memory.  Real
    # kernels, real control flow, no xGMI: the JSON line says so ("rehearsal": true) and is never a scaling measurement.
    rehearse = os.environ.get("MID_BENCH_REHEARSE") == "1"
    n_visible = max(torch.cuda.device_count(), 1)
    # (a launcher that narrows each rank's visibility to its own GPU -- ROCR/HIP_VISIBLE_DEVICES per rank -- leaves one visible
    # device, index 0, in every process: LOCAL_RANK is then not a device index)
    dev_index = local_rank % n_visible if (rehearse or local_rank >= n_visible) else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = torch.device("cpu") if rehearse else device        # where the small tensors of the collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        world = dist.get_world_size()           # n_gpus is what the process group says, not what the flags say

    import image_denoising_filter_amd as mid
    from image_denoising_filter_amd import sharding
    ctx = mid.Context(dev_index)
    # One explicit (non-default) stream for everything: torch ops, RCCL waits (req.wait() orders the CURRENT
    # torch stream), the library's kernels and the hipEvents that time them.  The default stream's handle is
    # 0, which the C-ABI reads as "use the context's own stream" -- that would not be ordered after RCCL.
    tstream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    process_group = None
    if world > 1:
        # what the ranks actually run on: the process group's backend and every rank's device, in the line itself
        names = [None] * world
        dist.all_gather_object(names, f"cuda:{dev_index} {ctx.name}")
        process_group = {"backend": str(dist.get_backend()), "devices": names,
                         "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}

    def barrier():
        if world > 1:
            dist.barrier()

    F = args.frames
    frames = synth_frames(F, 100 + rank, device)
    outs = [torch.empty((H, W, 4), device=device, dtype=torch.float32) for _ in range(F)]
    fptr, optr = [f.data_ptr() for f in frames], [o.data_ptr() for o in outs]

    if args.workload == "nlm":
        def step():
            ctx.nlm_temporal_dev(fptr, optr, W, H, HPARAM, SEARCH, PATCH, 0, 0, F, mid.FMT_RGBA32F, stream)
        flop_px, bytes_px, launches_per_step = NLM_FLOP_PER_PX, NLM_BYTES_PER_PX, 1
        metric = "Mpixel/s (NLM 21x21 search / 7x7 patch, 1920x1080 RGBA32F)"
        workload = "nlm_21x21_7x7_1080p_hdr (BASELINE configs[2]; single-frame NLM, fused accumulate+normalize)"
        kernel = "nlm_strip_kernel<-10,11,-3,4,8,4,f32,FUSED>"
        flop_note = "14,112/px (minimum-work separable NLM, SURVEY.md 8d)"
    else:
        def step():
            for i in range(F):
                ctx.bilateral_dev(fptr[i], optr[i], W, H, 8, 2.0, 0.2, mid.LAYOUT_LINEAR, mid.FMT_RGBA32F, stream)
        flop_px, bytes_px, launches_per_step = BIL_FLOP_PER_PX, BIL_BYTES_PER_PX, F
        metric = "Mpixel/s (bilateral r=8, 1920x1080 RGBA32F)"
        workload = "bilateral_r8_linear_1080p_hdr (BASELINE configs[1]; one launch per frame)"
        kernel = "bilateral_kernel<8,2,8,f32,LINEAR>"
        flop_note = "5,780/px (20 flop per tap x 17x17 taps, SURVEY.md 8d)"

    for _ in range(args.warmup):
        step()
    timers = Timers(mid, ctx, args.steps)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timers.tick(i, stream)
        step()
        timers.tock(i, stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = timers.ms()
    timers.close()
    avg_launch_s = sum(kernel_ms) / len(kernel_ms) / 1e3

    value = world * F * args.steps * NPIX / 1e6 / elapsed
    px_per_launch = F * NPIX // launches_per_step
    avg_launch_s /= launches_per_step        # the timers bracket one step = launches_per_step back-to-back launches
    fp = loaded_kernel_fingerprint(mid.LIB_PATH, args.workload)
    # (nlm: one launch covers F frames; bilateral: one frame per launch)
    traffic, traffic_source = load_traffic(F if args.workload == "nlm" else 1, fp, args.workload)
    res = {
        "metric": metric,
        "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        **({"rehearsal": True, "rehearsal_note": f"{world} ranks on {torch.cuda.device_count()} device(s), gloo + host-staged halo: control flow only, not a scaling measurement"} if rehearse else {}),
        "config": {"workload": workload,
                   "frames_per_gpu_per_step": F, "width": W, "height": H, "h": HPARAM,
                   "parallelism": f"frame-sharded x{world}, no data-path collective",
                   **({"process_group": process_group} if process_group else {})},
        "roofline": {
            "bound": "mfma", "bound_contract_side": "compute (the enum is hbm | mfma; the kernel issues no MFMA: bound_actual)", "bound_actual": "valu", "achieved": round(flop_px * px_per_launch / avg_launch_s / 1e12, 3),
            "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flop_px * px_per_launch / avg_launch_s / 1e12 / PEAK_FP32_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kernel, "avg_launch_ms": round(avg_launch_s * 1e3, 4),
            # spread of the K timed launches (hipEvent pairs on the launch stream, one pair per step)
            "kernel_ms_min": round(min(kernel_ms) / launches_per_step, 4), "kernel_ms_max": round(max(kernel_ms) / launches_per_step, 4),
            "kernel_code_sha256": fp[0],
            **load_utilisation(args.workload, fp),
            "note": "compute roofline: `bound` is the contract's two-valued enum (hbm | mfma) and this kernel belongs on its compute side; "
                    "`bound_actual` says what binds it in fact -- fp32 vector (VALU) issue: the kernel contains no MFMA instruction. The peak is "
                    "the fp32 vector peak, which equals the dense f32 MFMA peak on gfx950 (157.3 TFLOP/s). Algorithmic "
                    f"flops = {flop_note} x px per launch.",
            "hbm": {"achieved_GBs": round(bytes_px * px_per_launch / avg_launch_s / 1e9, 1),
                    "peak_GBs": PEAK_HBM_GBS,
                    "frac": round(bytes_px * px_per_launch / avg_launch_s / 1e9 / PEAK_HBM_GBS, 5)},
        },
    }

    # ---- outside the timed region -------------------------------------------------------------
    # BASELINE configs[2] read literally is ONE 1920x1080 frame per launch; `value` is the F-frame launch (whole rounds of
    # workgroups).  The single-frame figure goes into `config` next to it, measured in every run (also with --no-extras):
    # median of three bursts of 10 launches, hipEvents on the launch stream.
    if args.workload == "nlm":
        def burst(n=10):
            tm = Timers(mid, ctx, 1)
            tm.tick(0, stream)
            for _ in range(n):
                ctx.nlm_temporal_dev(fptr[:1], optr[:1], W, H, HPARAM, SEARCH, PATCH, 0, 0, 1, mid.FMT_RGBA32F, stream)
            tm.tock(0, stream)
            torch.cuda.synchronize()
            ms = tm.ms()[0] / n
            tm.close()
            return ms
        burst(3)
        s1 = sorted(burst() for _ in range(3))[1]
        res["config"]["single_frame_launch"] = {"ms": round(s1, 4), "Mpixel/s": round(NPIX / 1e3 / s1, 1),
                                                "frac_of_fp32_peak": round(NLM_FLOP_PER_PX * NPIX / (s1 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
                                                "what": "one 1920x1080 frame per launch (configs[2] read literally): 2 full rounds of workgroups + "
                                                        "the last round in the HALF shape; median of 3 bursts of 10 launches"}

    # Nothing below may cost the headline: every extra is individually guarded, and a watchdog prints the
    # line without the unfinished extras and ends every rank if they take longer than 5 minutes (a hung
    # collective would otherwise lose the whole run).
    import threading
    also = {}
    printed = threading.Lock()

    def emit():
        if printed.acquire(blocking=False) and rank == 0:
            res["also"] = also
            res.setdefault("cpu_baseline", None)
            print(json.dumps(res), file=json_out, flush=True)

    in_flight = {"extra": None}

    def on_timeout():
        also["error"] = (f"extras did not finish within {watchdog_s:g} s (in flight: {in_flight['extra']}); line emitted by the watchdog, "
                         "which then ends the process with exit code 3")
        emit()
        # A process that has touched the GPU and is abandoned in the middle of an extra -- possibly inside a collective
        # whose peers are still waiting -- must not report success, whatever the world size: the headline line above is
        # complete (the timed region and its max-over-ranks reduction finished on every rank before any extra started),
        # `also.error` names the extra that was in flight, and the exit code says that this run did not end cleanly.
        os._exit(3)
    watchdog_s = float(WATCHDOG_S)
    watchdog = threading.Timer(watchdog_s, on_timeout)
    watchdog.daemon = True
    watchdog.start()

    def guarded(name, fn):
        in_flight["extra"] = name
        try:
            fn()
        except Exception as e:          # an extra must never take the measurement down with it
            also[name + "_error"] = f"{type(e).__name__}: {e}"
        in_flight["extra"] = None

    XF = 16                             # frames of the 16-frame side measurements (bilateral batch, short pipelines), whatever --frames is
    if not args.no_extras:
        def time_gpu(fn, n=10):
            fn()
            torch.cuda.synchronize()
            tm = Timers(mid, ctx, 1)
            tm.tick(0, stream)
            for _ in range(n):
                fn()
            tm.tock(0, stream)
            torch.cuda.synchronize()
            ms = tm.ms()[0]
            tm.close()
            return ms / n / 1e3

        def extra_bilateral():
            # The two layouts run the same inner loop (identical opcode stream; only the tile fill differs).  Timed one after
            # the other, 10 launches each, whichever came SECOND read 3-6 % slower (rounds 1-2: 0.181 vs 0.193 ms; the
            # profiler's launch mix, which runs texture first, and an interleaved A/B both show them equal, LABNOTES.md rounds 1-3 section 3.2):
            # 2 ms of 0.18 ms launches between host-side gaps sit on the GPU's clock ramp.  So: warm both, then alternate
            # them (order flipped every repetition), 40 launches per timing (20 until round 6: 3 ms bursts still sat on the ramp and
            # hid a 3 % kernel gain that the 16-frame batch and a longer A/B showed), and report each layout's median.
            layouts = (("bilateral_r8_linear", mid.LAYOUT_LINEAR), ("bilateral_r8_texture", mid.LAYOUT_TEXTURE))
            fns = {name: (lambda lay=layout: ctx.bilateral_dev(fptr[0], optr[0], W, H, 8, 2.0, 0.2, lay, mid.FMT_RGBA32F, stream))
                   for name, layout in layouts}
            for name, _ in layouts:
                time_gpu(fns[name], 40)
            samples = {name: [] for name, _ in layouts}
            for rep in range(7):
                for name, _ in (layouts if rep % 2 == 0 else layouts[::-1]):
                    samples[name].append(time_gpu(fns[name], 40))
            for name, _ in layouts:
                s = sorted(samples[name])[len(samples[name]) // 2]
                also[name] = {"Mpixel/s": round(NPIX / 1e6 / s, 1), "ms": round(s * 1e3, 4), "ms_min": round(min(samples[name]) * 1e3, 4),
                              "valu_frac": round(BIL_FLOP_PER_PX * NPIX / s / 1e12 / PEAK_FP32_TFLOPS, 4),
                              "hbm_GBs": round(BIL_BYTES_PER_PX * NPIX / s / 1e9, 1),
                              "hbm_frac": round(BIL_BYTES_PER_PX * NPIX / s / 1e9 / PEAK_HBM_GBS, 5),
                              "timing": "median of 7 interleaved timings of 40 launches"}
            also["bilateral_r8_texture_over_linear"] = round(also["bilateral_r8_texture"]["ms"] / also["bilateral_r8_linear"]["ms"], 4)

        guarded("bilateral", extra_bilateral)

        def extra_bilateral_batch():
            # all F resident frames in ONE launch (mid_bilateral_batch): every round of workgroups is full
            nb = min(F, XF)
            s = time_gpu(lambda: ctx.bilateral_batch_dev(fptr[:nb], optr[:nb], W, H, 8, 2.0, 0.2, mid.LAYOUT_LINEAR, mid.FMT_RGBA32F, stream), 5)
            also["bilateral_r8_linear_batch"] = {"Mpixel/s": round(nb * NPIX / 1e6 / s, 1), "ms_per_frame": round(s * 1e3 / nb, 4), "frames": nb,
                                                 "valu_frac": round(BIL_FLOP_PER_PX * nb * NPIX / s / 1e12 / PEAK_FP32_TFLOPS, 4),
                                                 "hbm_GBs": round(BIL_BYTES_PER_PX * nb * NPIX / s / 1e9, 1)}

        guarded("bilateral_batch", extra_bilateral_batch)

        def extra_layers():
            # BASELINE configs[3]: 4 RGBA8 guide layers, fused layer-aware bilateral r=8 (16+4L+16 B/px)
            lay = [(f[..., :4].clamp(0, 1) * 255).to(torch.uint8).contiguous() for f in frames[:4]]
            tbl = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in lay])
            bp = mid.BilateralParams(W, H, 2.0, 0.2, 8, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)
            # Steady state: 60 untimed launches first.  After even a short idle gap (the torch ops above are enough) the first ~20
            # launches of this kernel run 15-25 % slower -- 0.87 -> 0.71 ms while the card's reported clock is at its highest and
            # board power is still climbing from 320 W towards 1300 W (tools/layers_spread.py, profiles/r04_layers_spread.txt) --
            # which is all the 21 % spread of round 3's ms_min_max was; in steady state p5..p95 is 0.700..0.710 ms and the time is
            # linear in the number of layers (issue-bound: not the LDS fit, not the launch tail).
            fn = lambda: mid.lib.mid_bilateral_layers(ctx.handle, ctypes.byref(bp), fptr[0], tbl, 4, optr[0], stream)   # noqa: E731
            time_gpu(fn, 60)
            ts_ = sorted(time_gpu(fn, 20) for _ in range(5))
            s = ts_[2]
            also["bilateral_layers_r8_L4_fused"] = {"Mpixel/s": round(NPIX / 1e6 / s, 1), "ms": round(s * 1e3, 4), "ms_min_max": [round(ts_[0] * 1e3, 4), round(ts_[-1] * 1e3, 4)],
                                                    "timing": "median of 5 timings of 20 launches after 60 untimed ones (steady state)",
                                                    "valu_frac": round(4 * BIL_FLOP_PER_PX * NPIX / s / 1e12 / PEAK_FP32_TFLOPS, 4),
                                                    "hbm_GBs": round(48 * NPIX / s / 1e9, 1)}

        guarded("layers", extra_layers)

        def extra_streaming():
            # streaming passes (HBM bound): normalize 48 B/px, pack 20 B/px, unpack 20 B/px
            # (buffers rotate over > 256 MiB so the Infinity Cache cannot serve the re-reads)
            wbufs = [torch.rand((H, W, 8), device=device, dtype=torch.float32) + 0.5 for _ in range(6)]
            rot = {"i": 0}

            def nxt(n):
                rot["i"] += 1
                return rot["i"] % n
            np_ = mid.NormalizeParams(W, H)
            s = time_gpu(lambda: mid.lib.mid_normalize(ctx.handle, ctypes.byref(np_), wbufs[nxt(6)].data_ptr(), optr[nxt(F)], stream), 24)
            also["normalize"] = {"ms": round(s * 1e3, 4), "hbm_GBs": round(48 * NPIX / s / 1e9, 1), "hbm_frac": round(48 * NPIX / s / 1e9 / PEAK_HBM_GBS, 4)}
            u8bufs = [torch.empty((H, W, 4), device=device, dtype=torch.uint8) for _ in range(8)]
            s = time_gpu(lambda: mid.lib.mid_pack_u8(ctx.handle, wbufs[nxt(6)].data_ptr(), NPIX * 4, u8bufs[nxt(8)].data_ptr(), stream), 24)
            also["pack_u8"] = {"ms": round(s * 1e3, 4), "hbm_GBs": round(20 * NPIX / s / 1e9, 1), "hbm_frac": round(20 * NPIX / s / 1e9 / PEAK_HBM_GBS, 4)}
            s = time_gpu(lambda: mid.lib.mid_unpack_u8(ctx.handle, u8bufs[nxt(8)].data_ptr(), NPIX * 4, 0, wbufs[nxt(6)].data_ptr(), stream), 24)
            also["unpack_u8"] = {"ms": round(s * 1e3, 4), "hbm_GBs": round(20 * NPIX / s / 1e9, 1), "hbm_frac": round(20 * NPIX / s / 1e9 / PEAK_HBM_GBS, 4)}

        guarded("streaming", extra_streaming)

        if "single_frame_launch" in res["config"]:
            also["nlm_single_frame_latency"] = {k_: res["config"]["single_frame_launch"][k_] for k_ in ("ms", "Mpixel/s")}

        def extra_temporal():
            # BASELINE configs[4]: ONE 64-frame sequence, temporal +-2 NLM, contiguous frame blocks over the ranks
            # (64/N frames each: strong scaling of a fixed job), halo frames from the neighbours over RCCL.
            k = 2
            n_seq = SEQ_FRAMES
            start, count = sharding.partition(n_seq, world)[rank]
            # frame g of the sequence has seed 1000+g on every rank count, so the job is the same at every N
            seq = [synth_frames(1, 1000 + g, device, shift=2 * g)[0] for g in range(start, start + count)]
            souts = [torch.empty((H, W, 4), device=device, dtype=torch.float32) for _ in range(count)]
            soptr = [o.data_ptr() for o in souts]
            stats, ev = {}, [torch.cuda.Event(enable_timing=True) for _ in range(2)]

            def launch(fr, first, cnt, off):
                ctx.nlm_temporal_dev([f.data_ptr() for f in fr], soptr[off:off + cnt], W, H, HPARAM, SEARCH, PATCH,
                                     k, first, cnt, mid.FMT_RGBA32F, stream)

            hooks = {"stats": stats, "before_wait": lambda: ev[0].record(tstream), "after_wait": lambda: ev[1].record(tstream)}

            def temporal_step():
                # halo isend/irecv posted first, interior frames filtered meanwhile, boundary frames after the wait
                sharding.temporal_block_overlapped(launch, seq, n_seq, k, hooks=hooks)

            temporal_step()
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            reps = 3
            held = 0.0
            for _ in range(reps):
                temporal_step()
                torch.cuda.synchronize()
                held += ev[0].elapsed_time(ev[1])
            barrier()
            te = (time.perf_counter() - t1) / reps
            per_rank = [te, held / reps, float(stats.get("halo_bytes_recv", 0)), float(stats.get("halo_bytes_sent", 0))]
            if world > 1:
                t = torch.tensor(per_rank, device=coll_device, dtype=torch.float64)
                allr = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(allr, t)
                rows = [[float(x) for x in a.tolist()] for a in allr]
            else:
                rows = [per_rank]
            te = max(r[0] for r in rows)
            also["temporal_nlm_k2"] = {"Mpixel/s_out": round(n_seq * NPIX / 1e6 / te, 1), "frames": n_seq,
                                       "frames_per_rank": [c for _, c in sharding.partition(n_seq, world)],
                                       "ms_per_sequence": round(te * 1e3, 3), "scaling": "strong (one 64-frame sequence)",
                                       "halo_bytes_recv_per_rank": [int(r[2]) for r in rows],
                                       "halo_bytes_sent_per_rank": [int(r[3]) for r in rows],
                                       "halo_held_ms_per_rank": [round(r[1], 4) for r in rows],
                                       "halo": ("RCCL isend/irecv of 2 frames per side, posted before the interior frames; held_ms = time "
                                                "the launch stream sat between the last interior launch and the first boundary launch")
                                               if world > 1 else "none (1 rank)"}

        guarded("temporal", extra_temporal)

        def extra_temporal_native():
            # The same job through the C++ path (csrc/sharded.cpp: mid_comm_create + mid_nlm_temporal_sharded -- RCCL bound by
            # dlopen, ncclSend/ncclRecv in one group on the library's exchange stream, interior launches meanwhile), with its
            # outputs compared bit for bit against a plain launch over block + halo frames fetched by the torch path.
            if rehearse:
                also["temporal_nlm_k2_native"] = {"skipped": "rehearsal: RCCL refuses two ranks on one device"}
                return
            k, n_seq = 2, SEQ_FRAMES
            start, count = sharding.partition(n_seq, world)[rank]
            seq = [synth_frames(1, 1000 + g, device, shift=2 * g)[0] for g in range(start, start + count)]
            souts = [torch.empty((H, W, 4), device=device, dtype=torch.float32) for _ in range(count)]
            idt = torch.zeros(mid.api.COMM_ID_BYTES, dtype=torch.uint8, device=coll_device)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(mid.comm_unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(idt, src=0)
            with mid.Comm(ctx, bytes(idt.cpu().numpy().tobytes()), rank, world) as comm:
                def step():
                    comm.nlm_temporal_sharded_dev([f.data_ptr() for f in seq], [o.data_ptr() for o in souts], W, H, n_seq, k,
                                                  HPARAM, SEARCH, PATCH, mid.FMT_RGBA32F, stream)
                step()
                torch.cuda.synchronize()
                # cross-check: the torch.distributed path's frames (own block + halo) through ONE plain launch
                have = sharding.exchange_halo(seq, n_seq, k)
                fr, first = sharding.window_for_block(have, n_seq, k, start, count)
                ref = [torch.empty((H, W, 4), device=device, dtype=torch.float32) for _ in range(count)]
                ctx.nlm_temporal_dev([f.data_ptr() for f in fr], [o.data_ptr() for o in ref], W, H, HPARAM, SEARCH, PATCH, k, first, count,
                                     mid.FMT_RGBA32F, stream)
                torch.cuda.synchronize()
                same = all(torch.equal(a, b) for a, b in zip(souts, ref))
                barrier()
                t1 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    step()
                    torch.cuda.synchronize()
                barrier()
                te = (time.perf_counter() - t1) / reps
                recv, sent, xms = comm.last_exchange()
                tl = comm.last_timeline()                       # device timeline of the LAST timed repetition
                order = comm.last_issue_order()
                rc_n, rc_rank, rc_ver = comm.rccl_info()        # what RCCL itself reports for this communicator
                prio = comm.stream_priority()
                bprio = comm.boundary_priority()
            hidden = tl["halo_hidden_frac"]
            per_rank = [te, float(recv), float(sent), float(xms), 1.0 if same else 0.0,
                        tl["exchange_start_ms"], tl["exchange_end_ms"], tl["interior_end_ms"], tl["end_ms"],
                        -1.0 if hidden is None else float(hidden), float(rc_n), float(rc_rank)]
            if world > 1:
                t = torch.tensor(per_rank, device=coll_device, dtype=torch.float64)
                allr = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(allr, t)
                rows = [[float(x) for x in a.tolist()] for a in allr]
            else:
                rows = [per_rank]
            nat, strong = native_temporal_report(rows, n_seq, world, rehearse, order, prio, rc_ver,
                                                 [c for _, c in sharding.partition(n_seq, world)])
            nat["boundary_stream_priority"] = list(bprio)       # the device's lowest: a pool of hardware queues of their own (LABNOTES R6.3)
            also["temporal_nlm_k2_native"] = nat
            res["scaling_strong"] = strong

        guarded("temporal_native", extra_temporal_native)

        def pcie_ceiling(nbytes_up, nbytes_down, n=16):
            """What the link gives in THIS run: pinned hipMemcpyAsync of n frames host->device on one stream while n frames go
            device->host on another (the pipeline's two copy directions), nothing else on the GPU.  GB/s per direction."""
            up, down = mid.PinnedFrames(ctx, n, nbytes_up), mid.PinnedFrames(ctx, n, nbytes_down)
            d_up, d_down = ctx.alloc(nbytes_up), ctx.alloc(nbytes_down)
            s_up, s_down = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
            try:
                def go():
                    for i in range(n):
                        assert mid.lib.mid_memcpy_h2d(ctx.handle, d_up.ptr, up.ptrs[i], nbytes_up, s_up.cuda_stream) == 0
                        assert mid.lib.mid_memcpy_d2h(ctx.handle, down.ptrs[i], d_down.ptr, nbytes_down, s_down.cuda_stream) == 0
                    ctx.sync(s_up.cuda_stream)
                    ctx.sync(s_down.cuda_stream)
                go()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    go()
                    ts.append(time.perf_counter() - t0)
                t = sorted(ts)[1]
                return {"h2d_GBs": round(n * nbytes_up / t / 1e9, 2), "d2h_GBs": round(n * nbytes_down / t / 1e9, 2),
                        "what": f"{n} pinned copies of {nbytes_up >> 20} MiB up and {nbytes_down >> 20} MiB down, concurrently on two streams, median of 3"}
            finally:
                up.free(); down.free(); d_up.free(); d_down.free()

        def pipeline_passes(fr, out_u8, passes=4, k=0):
            """The pipeline as a C caller sees it: frames already in pinned memory, a clock around the C call (and the call's own
            timings_ms[0] beside it -- since round 4 that covers the whole call too).  First call = the context's cache empty
            (mid_ctx_release_cached before it): ring, output slots and events are allocated inside the call.  Steady state =
            median of `passes` further calls, which allocate nothing."""
            h_, w_ = fr[0].shape[:2]
            fmt = mid.FMT_RGBA8 if fr[0].dtype == np.uint8 else mid.FMT_RGBA32F
            out_bytes = w_ * h_ * (4 if out_u8 else 16)
            # (frames repeat in the 64-frame sequences: one pinned buffer per distinct frame)
            uniq = {}
            for f in fr:
                uniq.setdefault(id(f), f)
            pin = mid.PinnedFrames(ctx, list(uniq.values()))
            ptr_of = dict(zip(uniq.keys(), pin.ptrs))
            hin = [ptr_of[id(f)] for f in fr]
            hout = mid.PinnedFrames(ctx, len(fr), out_bytes)
            try:
                def call():
                    t0 = time.perf_counter()
                    inside = ctx.sequence_nlm_pinned(hin, hout.ptrs, w_, h_, fmt, k=k, overlap=True, search=SEARCH, patch=PATCH, out_u8=out_u8)
                    return (time.perf_counter() - t0) * 1e3, inside
                ctx.release_cached()
                first_ms, first_in = call()
                rows = sorted((call() for _ in range(passes)), key=lambda r: r[0])
                wall, (wall_in, kern, copy) = rows[len(rows) // 2]
                mpx = lambda ms: round(len(fr) * NPIX / 1e3 / ms, 1)       # noqa: E731
                return {"Mpixel/s_overlap": mpx(wall), "Mpixel/s_first_call": mpx(first_ms), "frames": len(fr),
                        "call_ms": round(wall, 3), "call_ms_inside": round(wall_in, 3), "first_call_ms": round(first_ms, 3),
                        "first_call_ms_inside": round(first_in[0], 3), "kernel_ms": round(kern, 3), "copy_ms": round(copy, 3),
                        f"Mpixel/s_min_max_of_{passes}_passes": [mpx(rows[-1][0]), mpx(rows[0][0])],
                        "timing": f"clock around the C call, frames already pinned; first call with the context's cache released, then median of {passes} calls"}
            finally:
                pin.free(); hout.free()

        def extra_pipeline():
            if rank == 0 and world == 1:
                # PCIe-inclusive: pinned host frames in, host frames out, overlapped streams (never `value`)
                hf = [f.cpu().numpy() for f in frames[:XF]]
                ceil32 = pcie_ceiling(NPIX * 16, NPIX * 16)
                r = pipeline_passes(hf, out_u8=False)
                _, (wall0, _, _) = ctx.sequence_nlm(hf, k=0, overlap=False, search=SEARCH, patch=PATCH)
                r["Mpixel/s_serial"] = round(len(hf) * NPIX / 1e3 / wall0, 1)
                r["pcie_ceiling"] = ceil32
                # each direction moves 16 B per pixel; the slower direction of the concurrent-copy ceiling bounds the frame rate
                r["pcie_frac"] = round(r["Mpixel/s_overlap"] * 16e6 / 1e9 / min(ceil32["h2d_GBs"], ceil32["d2h_GBs"]), 4)
                r["note"] = ("host RGBA32F frames in pinned memory -> H2D, NLM (two alternating kernel streams), D2H, all overlapped; "
                             "serial = a sync after every step like the reference's fence; pcie_frac = 16 B/px each way against the "
                             "slower direction of the concurrent pinned-copy ceiling measured in this run")
                also["pipeline_pcie_inclusive"] = r
                # the reference's LDR path: RGBA8 frames in, RGBA8 frames out (u8 conversion on the device)
                lf = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in hf]
                r8 = pipeline_passes(lf, out_u8=True)
                r8["note"] = ("host RGBA8 frames in, RGBA8 frames out (mid_sequence_nlm_range_u8), 4 B/px each way.  Round 6: the outputs are in "
                              "page-locked memory, so the kernel's epilogue stores them straight into the caller's buffers -- no download stage "
                              "(copy_ms = uploads only).  Bound by the kernel streams: two co-running single-frame launches, kernel_ms / 2 of "
                              "call_ms busy per stream; the staged download it replaced was bound by the runtime's device-to-host copies turning "
                              "3.5x slower part way into a 64-frame call (profiles/r06_pipeline_u8_timeline.txt, LABNOTES R6.1)")
                r8["pcie_frac"] = round(r8["Mpixel/s_overlap"] * 4e6 / 1e9 / min(ceil32["h2d_GBs"], ceil32["d2h_GBs"]), 4)
                also["pipeline_pcie_inclusive_ldr"] = r8

        guarded("pipeline", extra_pipeline)

        def extra_pipeline_long():
            if rank == 0 and world == 1:
                # the same pipeline over a 64-frame sequence (the length of BASELINE configs[4]): fill, drain and the GPU's
                # clock ramp after idle (about 2.8 ms per cold start, tools/pipe_idle_ab.py) weigh a quarter as much
                # exactly SEQ_FRAMES frames whatever --frames is: the first XF resident frames cycled (the keys say _64)
                hf = [f.cpu().numpy() for f in frames[:XF]]
                lf8 = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in hf]
                # (six passes: a pass that follows a host-side gap runs 8 % slower on the device's clocks, profiles/r06_pipe_u8_spread.txt)
                also["pipeline_pcie_inclusive_ldr_64"] = pipeline_passes([lf8[i % len(lf8)] for i in range(SEQ_FRAMES)], out_u8=True, passes=6)
                r = pipeline_passes([hf[i % len(hf)] for i in range(SEQ_FRAMES)], out_u8=False)
                c = also.get("pipeline_pcie_inclusive", {}).get("pcie_ceiling")
                if c:
                    r["pcie_frac"] = round(r["Mpixel/s_overlap"] * 16e6 / 1e9 / min(c["h2d_GBs"], c["d2h_GBs"]), 4)
                also["pipeline_pcie_inclusive_64"] = r

        guarded("pipeline_long", extra_pipeline_long)

        def extra_pipeline_k2():
            if rank == 0 and world == 1:
                # BASELINE configs[4] host -> host on ONE GPU: mid_sequence_nlm with k = 2 over a 64-frame 1080p RGBA32F sequence in
                # pinned memory -- the reference's "async copy overlap" mode (RecordCommandsOfOverlappingNLM, src/main.cpp:889-989,
                # loop :1539-1573) with the explicit +-2 window.  Five frame pairs per output make it kernel-bound: the copies (16 B/px
                # each way) hide behind the launches; the figure to hold it against is temporal_nlm_k2 (frames resident in HBM).
                hf = [f.cpu().numpy() for f in frames[:XF]]
                r = pipeline_passes([hf[i % len(hf)] for i in range(SEQ_FRAMES)], out_u8=False, passes=2, k=2)
                r = {("out-" + kk if kk.startswith("Mpixel/s") else kk): v for kk, v in r.items()}
                res_t = also.get("temporal_nlm_k2", {}).get("Mpixel/s_out")
                if res_t:
                    r["ratio_to_temporal_nlm_k2"] = round(r["out-Mpixel/s_overlap"] / res_t, 4)
                r["k"] = 2
                r["note"] = ("64 x 1080p RGBA32F host frames in pinned memory -> H2D, temporal NLM +-2 (one output per launch, two alternating "
                             "kernel streams, ring of 2k+4 device frames), D2H; output-Mpixel/s; kernel_ms = sum over both kernel streams, "
                             "copy_ms = sum of uploads and downloads; ratio_to_temporal_nlm_k2 = against the same sequence resident in HBM")
                also["pipeline_k2_64"] = r

        guarded("pipeline_k2", extra_pipeline_k2)

        def extra_multiframe():
            if rank == 0 and world == 1:
                # The reference's literal multi-frame mode with copy/compute overlap (mid_nlm_multiframe = RecordCommandsOfOverlappingNLM,
                # src/main.cpp:889-989, loop :1539-1573: ONE target, nine neighbour frames streamed from the host, then normalize), with
                # the frames in page-locked memory and in ordinary (pageable) memory -- the latter is what an integrator who keeps
                # std::vector frames gets; the library bounces those through its own pinned buffers (csrc/hostcopy.cpp).
                nfr = 9
                hf = [frames[i % len(frames)].cpu().numpy() for i in range(nfr)]      # (--frames may be smaller than nine)
                prm = mid.NlmParams(W, H, HPARAM, SEARCH[0], SEARCH[1], PATCH[0], PATCH[1], mid.FMT_RGBA32F)
                pin = mid.PinnedFrames(ctx, hf)
                pout = mid.PinnedFrames(ctx, 1, NPIX * 16)
                pageable_out = np.empty((H, W, 4), np.float32)
                try:
                    def call(ptrs, target, out_ptr):
                        t = (ctypes.c_float * 3)()
                        t0 = time.perf_counter()
                        rc = mid.lib.mid_nlm_multiframe(ctx.handle, ctypes.byref(prm), target, (ctypes.c_void_p * nfr)(*ptrs), nfr, out_ptr, 1, t)
                        assert rc == 0, mid.lib.mid_last_error()
                        return (time.perf_counter() - t0) * 1e3, t[1], t[2]
                    res_m = {}
                    for name, ptrs, out_ptr in (("pinned", list(pin.ptrs), pout.ptrs[0]),
                                                ("pageable", [f.ctypes.data for f in hf], pageable_out.ctypes.data)):
                        call(ptrs, ptrs[0], out_ptr)
                        rows = sorted(call(ptrs, ptrs[0], out_ptr) for _ in range(3))
                        wall, kern, copy = rows[1]
                        res_m[name] = {"call_ms": round(wall, 3), "kernel_ms": round(kern, 3), "copy_ms": round(copy, 3)}
                    same = np.array_equal(pout.array(0, (H, W, 4), np.float32), pageable_out)
                    res_m["outputs_equal"] = bool(same)
                    res_m["note"] = ("one 1080p RGBA32F target accumulated over 9 host frames (unfused mid_nlm_accum per frame + normalize, the reference's own "
                                     "schedule) with the next frames' uploads overlapped; median of 3 calls; pageable = frames and result in ordinary memory")
                    also["nlm_multiframe_overlap_9"] = res_m
                finally:
                    pin.free(); pout.free()

        guarded("multiframe", extra_multiframe)

        def extra_graph():
            if rank == 0 and world == 1:
                # The reference records its dispatches once into a Vulkan command buffer and submits the recording
                # (RecordCommandsOfExecuteNLM src/main.cpp:849-887, RunCommandBuffer :1078-1103); the HIP counterpart is a captured
                # graph, offered as mid_record_begin / mid_record_end / mid_recording_submit (csrc/recording.cpp,
                # tests/test_gpu_z_recording.py).  Does a recording pay?  The literal multi-frame mode -- clear, nine nonlocal.comp
                # dispatches at the reference's window, normalize: 11 launches -- issued call by call against one submission, on small
                # frames and on 1080p (LABNOTES R6.10: it does not; the line keeps the evidence).
                out = {}
                for (hh, ww, reps) in ((128, 128, 200), (256, 256, 200), (H, W, 10)):
                    fr = [torch.rand((hh, ww, 4), device=device, dtype=torch.float32) for _ in range(9)]
                    for f in fr:
                        f[..., 3] = 1.0
                    Wb = torch.empty((hh, ww, 8), device=device, dtype=torch.float32)
                    o = torch.empty((hh, ww, 4), device=device, dtype=torch.float32)
                    prm = mid.NlmParams(ww, hh, HPARAM, -7, 7, -3, 3, mid.FMT_RGBA32F)
                    pn = mid.NormalizeParams(ww, hh)
                    gs = torch.cuda.Stream()

                    def body(st):
                        rc = mid.lib.mid_memset(ctx.handle, Wb.data_ptr(), 0, Wb.numel() * 4, st)
                        assert rc == 0, mid.lib.mid_last_error()
                        for f in fr:
                            rc = mid.lib.mid_nlm_accum(ctx.handle, ctypes.byref(prm), fr[4].data_ptr(), f.data_ptr(), Wb.data_ptr(), st)
                            assert rc == 0, mid.lib.mid_last_error()
                        rc = mid.lib.mid_normalize(ctx.handle, ctypes.byref(pn), Wb.data_ptr(), o.data_ptr(), st)
                        assert rc == 0, mid.lib.mid_last_error()
                    torch.cuda.synchronize()
                    body(gs.cuda_stream)
                    gs.synchronize()
                    want = o.clone()
                    with ctx.record(gs.cuda_stream) as rec:            # mid_record_begin ... mid_record_end (csrc/recording.cpp)
                        body(gs.cuda_stream)
                    o.zero_()
                    torch.cuda.synchronize()
                    rec.submit(gs.cuda_stream)
                    torch.cuda.synchronize()
                    same = bool(torch.equal(o, want))

                    def timed(fn):
                        fn(); torch.cuda.synchronize()
                        ts = []
                        for _ in range(3):
                            t0 = time.perf_counter()
                            for _ in range(reps):
                                fn()
                            torch.cuda.synchronize()
                            ts.append((time.perf_counter() - t0) / reps * 1e3)
                        return sorted(ts)[1]

                    t_call, t_graph = timed(lambda: body(gs.cuda_stream)), timed(lambda: rec.submit(gs.cuda_stream))
                    out[f"{ww}x{hh}"] = {"by_call_ms": round(t_call, 4), "recording_submit_ms": round(t_graph, 4),
                                         "ratio": round(t_call / t_graph, 3), "outputs_equal": same, "graph_nodes": rec.info()[0]}
                    rec.close()
                out["note"] = ("the reference's literal multi-frame sequence (1 memset + 9 mid_nlm_accum at [-7,7)/[-3,3) + mid_normalize) per target frame, issued "
                               "through the C-ABI call by call (ctypes) against one mid_recording_submit of the recorded calls (a captured hipGraph: the counterpart of "
                               "the reference's recorded command buffers); median of 3 timed loops.  Same bytes, NO gain at any size: the nine accumulates depend on each other and a "
                               "tile's 196 offsets take ~45 us whatever the frame, so the chain is 0.46 ms from 64x64 to 512x512 (the fused mid_nlm_temporal is the answer "
                               "to that, not a graph); from compiled host code, also for 28 short launches (4 us per launch either way): profiles/r06_recording_replay.txt")
                also["graph_replay_literal_nlm"] = out

        guarded("graph", extra_graph)

        def extra_image_io():
            if rank == 0 and world == 1:
                # what the drop-in COMMAND spends per file around the GPU work -- the library's own PNG / EXR codecs
                # (mid_image_save / mid_image_load, host only; the reference calls lodepng / tinyexr, src/main.cpp:155,196,1699,1717)
                # on one 1080p frame of the bench's data, a call's internal parallelism at its default (up to 16 host threads).
                import tempfile
                f32 = frames[0].cpu().numpy()
                u8 = np.clip(f32 * 64.0, 0, 255).astype(np.uint8)
                out = {}
                with tempfile.TemporaryDirectory() as d:
                    for name, arr in (("png", u8), ("exr", f32)):
                        path = os.path.join(d, "f." + name)
                        enc, dec = [], []
                        for _ in range(3):
                            t0 = time.perf_counter(); mid.save_image(path, arr); enc.append(time.perf_counter() - t0)
                            t0 = time.perf_counter(); back = mid.load_image(path); dec.append(time.perf_counter() - t0)
                        assert np.array_equal(back, arr)
                        out[name] = {"encode_ms": round(sorted(enc)[1] * 1e3, 1), "decode_ms": round(sorted(dec)[1] * 1e3, 1),
                                     "file_MB": round(os.path.getsize(path) / 1e6, 2)}
                out["note"] = ("one 1920x1080 frame per call, median of 3, host threads as the library picks them (min(16, hardware threads)); a PNG's inflate "
                               "is one serial stream, which is why mi_denoise takes the files of a sequence one per worker thread "
                               "(profiles/r06_cli_animation_time.txt: 64 files 7.2 -> 2.45 s end to end, of which the GPU 0.04 s)")
                also["image_io_1080p"] = out

        guarded("image_io", extra_image_io)

    res["also"] = also

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(args.workload)
        except Exception as e:  # the baseline is a reported extra; never fail the GPU measurement on it
            res["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    elif rank == 0:
        res["cpu_baseline"] = None

    watchdog.cancel()
    emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
