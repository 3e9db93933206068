"""Workload for `rocprofv3 --kernel-trace`: the host->host frame pipeline (event-joined default) over 64 x 1080p RGBA8
frames, 21x21/7x7, k=0, run twice (the second pass is the one summarised).  Usage on the GPU box:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_pipe -- python3 $R/tools/pipeline_trace.py
then  python tools/pipeline_trace.py --summarise gpurun_out/prof_pipe  ->  profiles/r02_pipeline_occupancy.json

Round 5: with `--kernel-trace --memory-copy-trace --marker-trace` (trace domains only, no counters) and
    python tools/pipeline_trace.py --summarise-stages gpurun_out/prof_pipe_r05 r05 [f32]
the library's ROCTx ranges (csrc/markers.cpp: one range per pipeline call, "upload f" / "nlm t" / "download t" inside) are
lined up with the copy and kernel records of the same run -> profiles/<tag>_pipeline_occupancy.json: per-stage occupancy of the
DEVICE (share of the call's span with an upload / a kernel / a download in flight) and of the HOST thread (share of the span
spent issuing each stage).  `f32` as the workload argument streams RGBA32F frames (link-bound) instead of RGBA8 (kernel-bound)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    tr = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))]
    nlm = sorted((a, b) for a, b, n in rows if "nlm_strip_kernel" in n and b - a > 300000)     # the 1080p launches (not the 2-frame warm-up's tiny ones)
    nlm = nlm[-64:]                                                                            # the second 64-frame pass
    span = nlm[-1][1] - nlm[0][0]
    busy, cur_a, cur_b, two = 0, nlm[0][0], nlm[0][1], 0
    for a, b in nlm[1:]:
        if a <= cur_b:
            two += min(b, cur_b) - a                                                           # time with (at least) two launches in flight
            cur_b = max(cur_b, b)
        else:
            busy += cur_b - cur_a
            cur_a, cur_b = a, b
    busy += cur_b - cur_a
    blit = [(a, b) for a, b, n in rows if "copyBuffer" in n and a >= nlm[0][0] and b <= nlm[-1][1]]
    out = {"command": "rocprofv3 --kernel-trace -- python3 tools/pipeline_trace.py (64 x 1080p RGBA8 in/out, 21x21/7x7, k=0, event-joined pipeline, second pass)",
           "launches": len(nlm), "span_ms": span / 1e6, "ms_per_frame": span / 1e6 / len(nlm),
           "nlm_kernel_in_flight_frac": busy / span, "two_or_more_launches_in_flight_frac": two / span,
           "avg_launch_ms": sum(b - a for a, b in nlm) / len(nlm) / 1e6,
           "largest_gap_without_an_nlm_launch_us": max([0] + [(nlm[i + 1][0] - max(x[1] for x in nlm[:i + 1])) / 1e3 for i in range(len(nlm) - 1)]),
           "blit_copy_kernels_inside_span": len(blit), "blit_copy_ms_total": sum(b - a for a, b in blit) / 1e6,
           "Mpixel/s_over_span": len(nlm) * 1920 * 1080 / (span / 1e9) / 1e6}
    dst = os.path.join(ROOT, "profiles", "r02_pipeline_occupancy.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise-stages":
    src, tag = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "r05")
    host_src = sys.argv[5] if len(sys.argv) > 5 else None      # a `--marker-trace` ONLY pass: the host-side shares with the least tracer overhead

    def newest(pat):
        c = sorted(glob.glob(os.path.join(src, "**", pat), recursive=True), key=os.path.getmtime)
        return c[-1] if c else None

    def col(row, *names):
        for n in names:
            if n in row:
                return row[n]
        raise KeyError(f"none of {names} in {list(row)}")

    def union(iv):
        iv = sorted(iv)
        tot, a, b = 0, None, None
        for x, y in iv:
            if a is None:
                a, b = x, y
            elif x <= b:
                b = max(b, y)
            else:
                tot += b - a
                a, b = x, y
        return tot + (b - a if a is not None else 0)

    def host_shares(path):
        marks = [(col(r, "Function", "Message", "Name"), int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))) for r in csv.DictReader(open(path))]
        calls = [m for m in marks if m[0].startswith("mid_sequence_nlm")]
        if not calls:
            sys.exit("no mid_sequence_nlm range in " + path)
        name, t0, t1 = calls[-1]                                               # the last (steady-state) pipeline call
        span = t1 - t0
        inside = [m for m in marks if m[1] >= t0 and m[2] <= t1 and m is not calls[-1]]
        host = {"span_ms": round(span / 1e6, 3)}
        for stage in ("upload", "nlm", "download", "drain"):
            iv = [(a, b) for n, a, b in inside if n.split()[0] == stage]
            host[stage] = {"ranges": len(iv), "host_ms": round(sum(b - a for a, b in iv) / 1e6, 3), "share_of_call": round(sum(b - a for a, b in iv) / span, 4)}
        return name, t0, t1, host

    mk = newest("*marker_api_trace.csv")
    name, t0, t1, host = host_shares(mk)
    span = t1 - t0
    host_source = mk
    if host_src:
        c = sorted(glob.glob(os.path.join(host_src, "**", "*marker_api_trace.csv"), recursive=True), key=os.path.getmtime)
        if c:
            _, _, _, host = host_shares(c[-1])
            host_source = c[-1]
    kt = newest("*kernel_trace.csv")
    kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))]
    nlm = [(a, b) for a, b, n in kern if "nlm_strip_kernel" in n and a >= t0 and b <= t1]
    blit = [(a, b) for a, b, n in kern if "copyBuffer" in n and a >= t0 and b <= t1]
    mc = newest("*memory_copy_trace.csv")
    up, down = [], []
    if mc:
        for r in csv.DictReader(open(mc)):
            a, b = int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))
            if a < t0 or b > t1:
                continue
            d = col(r, "Direction", "Name", "Kind").upper()
            (up if ("HOST_TO_DEVICE" in d or "H2D" in d or "HOSTTODEVICE" in d) else down if ("DEVICE_TO_HOST" in d or "D2H" in d or "DEVICETOHOST" in d) else []).append((a, b))
    n_out = host["nlm"]["ranges"]
    out = {"command": "rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -- python3 tools/pipeline_trace.py " + " ".join(sys.argv[4:5]),
           "call": name, "span_ms": round(span / 1e6, 3), "outputs": n_out,
           "Mpixel/s_over_span": round(n_out * 1920 * 1080 / (span / 1e9) / 1e6, 1) if n_out else None,
           "device_stage_occupancy": {
               "upload_copies": {"records": len(up), "busy_ms": round(union(up) / 1e6, 3), "share_of_call": round(union(up) / span, 4)},
               "nlm_kernels": {"records": len(nlm), "busy_ms": round(union(nlm) / 1e6, 3), "share_of_call": round(union(nlm) / span, 4),
                               "sum_of_launch_ms": round(sum(b - a for a, b in nlm) / 1e6, 3)},
               "download_copies": {"records": len(down), "busy_ms": round(union(down) / 1e6, 3), "share_of_call": round(union(down) / span, 4)},
               "blit_copy_kernels": {"records": len(blit), "busy_ms": round(union(blit) / 1e6, 3), "share_of_call": round(union(blit) / span, 4)},
               "def": "busy = union of the records' [start, end] intervals inside the call's ROCTx range; memory-copy records are the runtime's SDMA/blit "
                      "copies (direction from the trace), kernels by name"},
           "host_stage_occupancy": dict(host, **{"source": os.path.relpath(host_source, ROOT),
                                                 "def": "time the calling thread spent inside the library's 'upload f' / 'nlm t' / 'download t' / 'drain' ranges "
                                                        "(issuing work, or -- drain -- waiting for it), as shares of THIS pass's span_ms; taken from a --marker-trace "
                                                        "only pass when one is given, because every traced domain slows the issuing thread"}),
           "tracer_overhead": "the same call takes about 50 ms untraced, 60 ms with --marker-trace alone and 70-95 ms with kernel + memory-copy tracing on top "
                              "(gpurun_out/prof_r05/pipe_f32_*.log): under a tracer the HOST thread sets the pace, untraced the link does (bench.py pcie_frac 0.95)",
           "sources": [os.path.relpath(x, ROOT) for x in (mk, kt, mc) if x]}
    dst = os.path.join(ROOT, "profiles", f"{tag}_pipeline_occupancy.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)

import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
f32 = len(sys.argv) > 1 and sys.argv[1] == "f32"
if f32:
    frames = [(rng.random((1080, 1920, 4), dtype=np.float32) * 4).astype(np.float32) for _ in range(16)] * 4
else:
    frames = [rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8) for _ in range(16)] * 4
ctx.sequence_nlm(frames[:2], k=0, out_u8=not f32, **mid.NLM_BENCH)
for rep in range(2):
    _, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, overlap=True, out_u8=not f32, **mid.NLM_BENCH)
    print(f"pass {rep}: wall {wall:.2f} ms -> {64 * 1920 * 1080 / wall / 1e3:.0f} Mpixel/s")
