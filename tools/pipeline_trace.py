"""Workload for `rocprofv3 --kernel-trace`: the host->host frame pipeline (event-joined default) over 64 x 1080p RGBA8
frames, 21x21/7x7, k=0, run twice (the second pass is the one summarised).  Usage on the GPU box:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_pipe -- python3 $R/tools/pipeline_trace.py
then  python tools/pipeline_trace.py --summarise gpurun_out/prof_pipe  ->  profiles/r02_pipeline_occupancy.json"""
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

import numpy as np
import image_denoising_filter_amd as mid
ctx = mid.Context(0)
rng = np.random.default_rng(0)
frames = [rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8) for _ in range(16)] * 4
ctx.sequence_nlm(frames[:2], k=0, out_u8=True, **mid.NLM_BENCH)
for rep in range(2):
    _, (wall, kern, copy) = ctx.sequence_nlm(frames, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
    print(f"pass {rep}: wall {wall:.2f} ms -> {64 * 1920 * 1080 / wall / 1e3:.0f} Mpixel/s")
