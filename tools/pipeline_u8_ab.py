"""What binds the RGBA8 (PNG in, PNG out) frame pipeline?  (the reference's default path, src/main.cpp:1945,
overlap recorder :889-989, loop :1539-1573.)

One child process per variant (the runtime reads GPU_MAX_HW_QUEUES when HIP initialises; the library's experiment knobs are read
when a context is created).  Each child runs mid_sequence_nlm_range_u8 over 64 (and 16) pinned 1080p RGBA8 frames, 21x21/7x7,
k=0 -- bench.py's also.pipeline_pcie_inclusive_ldr_64 -- times PASSES calls with a clock around the C call, hashes the outputs
(every variant must give the same bytes) and reads the DEVICE timeline of the last call back from the library's own events
(mid_pipe_last_timeline: no profiler, so the call runs at its own pace).

  python tools/pipeline_u8_ab.py                 # driver: all variants, one after the other, A B A B order
  python tools/pipeline_u8_ab.py --child NAME    # one variant in this process (environment already set)
  python tools/pipeline_u8_ab.py --timeline FILE # child: also write the per-frame timeline table to FILE
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# The library-side switches (MID_PIPE_EXP=depth=8 | pace=N | direct=1 | copyprio=-1) existed only while this A/B was run; what they
# were is recorded in tools/experiments/README.md (round 6) and the results in profiles/r06_pipeline_u8_ab.txt.  `direct` is now what
# the library does for RGBA8 outputs in pinned memory; against the current tree only the environment variants mean anything.
VARIANTS = [
    ("shipped", {}),
    ("hwq8", {"GPU_MAX_HW_QUEUES": "8"}),
    ("copyprio_high", {"MID_PIPE_EXP": "copyprio=-1"}),
    ("eager", {"MID_PIPE_EXP": "eager"}),
    ("eager+copyprio_high", {"MID_PIPE_EXP": "eager,copyprio=-1"}),
    ("eager+hwq8", {"MID_PIPE_EXP": "eager", "GPU_MAX_HW_QUEUES": "8"}),
    ("pace3", {"MID_PIPE_EXP": "pace=3"}), ("pace4", {"MID_PIPE_EXP": "pace=4"}), ("pace6", {"MID_PIPE_EXP": "pace=6"}),
    ("pace8", {"MID_PIPE_EXP": "pace=8"}), ("pace12", {"MID_PIPE_EXP": "pace=12"}),
    ("pace6+depth8", {"MID_PIPE_EXP": "pace=6,depth=8"}), ("pace4+direct", {"MID_PIPE_EXP": "pace=4,direct=1"}),
    ("depth8", {"MID_PIPE_EXP": "depth=8"}), ("direct", {"MID_PIPE_EXP": "direct=1"}),
    ("direct+hwq8", {"MID_PIPE_EXP": "direct=1", "GPU_MAX_HW_QUEUES": "8"}),
    ("copyprio_high+depth8", {"MID_PIPE_EXP": "copyprio=-1,depth=8"}),
    ("hwq8+depth8", {"MID_PIPE_EXP": "depth=8", "GPU_MAX_HW_QUEUES": "8"}),
]


def analyse(up, out, depth, label):
    """Per output j: what its kernel start waited for.  Candidates: its frame's upload end, the previous launch on ITS stream
    (j-2), the output slot (download j-depth end).  gate = the latest of them; lag = kernel start - gate."""
    up_end = {f: e for f, _, e in up}
    rows, idle = [], {"upload": 0.0, "slot": 0.0, "stream": 0.0, "start": 0.0}
    lines = [f"# {label}", "# frame | upload start..end | kernel start..end (stream) | download start..end | gated by | lag after gate (ms)"]
    for j, (f, c0, c1, d0, d1) in enumerate(out):
        cands = {"upload": up_end.get(f, 0.0)}
        if j >= 2:
            cands["stream"] = out[j - 2][2]
        if depth and j >= depth:
            cands["slot"] = out[j - depth][4]
        gate = max(cands, key=cands.get)
        lag = c0 - cands[gate]
        prev_end = out[j - 2][2] if j >= 2 else 0.0
        # time this launch's stream sat idle before it, attributed to the gating dependency
        idle["start" if j < 2 else gate] += max(0.0, c0 - prev_end)
        u = next((x for x in up if x[0] == f), (f, float("nan"), float("nan")))
        lines.append(f"{f:3d} | {u[1]:7.3f}..{u[2]:7.3f} | {c0:7.3f}..{c1:7.3f} (s{j & 1}) | {d0:7.3f}..{d1:7.3f} | {gate:6s} | {lag:6.3f}")
        rows.append((f, c0, c1, d0, d1, gate, lag))
    n = len(out)
    span = max(r[4] for r in rows)
    kern = sum(r[2] - r[1] for r in rows)
    copies = [(e - s) for _, s, e in up] + [(r[4] - r[3]) for r in rows]
    summ = {"frames": n, "span_ms": round(span, 3), "ms_per_frame": round(span / n, 4), "kernel_sum_ms": round(kern, 3),
            "avg_kernel_ms": round(kern / n, 4), "kernel_stream_busy_frac": round(kern / (2 * span), 4),
            "avg_upload_ms": round(sum(e - s for _, s, e in up) / len(up), 4),
            "avg_download_ms": round(sum(r[4] - r[3] for r in rows) / n, 4),
            "max_copy_ms": round(max(copies), 3),
            "kernel_stream_idle_ms_by_gate": {k: round(v, 3) for k, v in idle.items()},
            "launches_gated_by": {g: sum(1 for r in rows if r[5] == g) for g in ("upload", "stream", "slot")},
            "avg_lag_after_gate_ms": round(sum(r[6] for r in rows) / n, 4)}
    return lines, summ


def child(name, timeline_file):
    import numpy as np
    import torch
    import image_denoising_filter_amd as mid
    import bench
    dev = torch.device("cuda", 0)
    ctx = mid.Context(0)
    fr = [f.cpu().numpy() for f in bench.synth_frames(16, 100, dev)]
    lf = [np.clip(f * 64.0, 0, 255).astype(np.uint8) for f in fr]
    W, H = 1920, 1080
    pin = mid.PinnedFrames(ctx, lf)
    res = {"variant": name, "env": {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "MID_PIPE_EXP") if os.environ.get(k)}}
    exp = os.environ.get("MID_PIPE_EXP", "")
    depth = 0 if "direct=1" in exp else (8 if "depth=8" in exp else 4)
    for nfr, passes in (((16, 6), (64, 8)) if not os.environ.get("MID_AB_SHAPES") else
                        [tuple(int(v) for v in x.split("x")) for x in os.environ["MID_AB_SHAPES"].split(",")]):
        hin = [pin.ptrs[i % 16] for i in range(nfr)]
        hout = mid.PinnedFrames(ctx, nfr, W * H * 4)

        def call():
            t0 = time.perf_counter()
            t = ctx.sequence_nlm_pinned(hin, hout.ptrs, W, H, mid.FMT_RGBA8, k=0, overlap=True, out_u8=True, **mid.NLM_BENCH)
            return (time.perf_counter() - t0) * 1e3, t
        call(); call()
        if os.environ.get("MID_AB_MARK"):
            print("MARK timed passes start", file=sys.stderr, flush=True)
        walls = []
        for _ in range(passes):
            w, t = call()
            walls.append((w, t[1], t[2]))
        h = hashlib.sha256()
        for i in range(nfr):
            h.update(hout.array(i, (H, W, 4), np.uint8).tobytes())
        ws = sorted(x[0] for x in walls)
        mpx = lambda ms: round(nfr * W * H / 1e3 / ms, 1)   # noqa: E731
        up, out = ctx.pipe_last_timeline()
        lines, summ = analyse(up, out, depth, f"{name}: {nfr} frames, last of {passes} passes (wall {walls[-1][0]:.2f} ms)")
        res[f"frames_{nfr}"] = {"Mpixel/s_median": mpx(ws[len(ws) // 2]), "Mpixel/s_min_max": [mpx(ws[-1]), mpx(ws[0])],
                                "spread_pct": round(100 * (ws[-1] - ws[0]) / ws[len(ws) // 2], 1),
                                "wall_ms": [round(x, 2) for x in ws], "kernel_sum_ms": round(walls[-1][1], 2),
                                "copy_sum_ms": round(walls[-1][2], 2), "sha256_of_outputs": h.hexdigest()[:16], "timeline": summ}
        if timeline_file and nfr == 64:
            with open(timeline_file, "w") as f:
                f.write("\n".join(lines) + "\n# summary: " + json.dumps(summ) + "\n")
        hout.free()
    pin.free()
    if os.environ.get("MID_AB_F32"):
        # the link-bound RGBA32F pipeline over 64 frames, for switches that touch the copy streams
        pin32 = mid.PinnedFrames(ctx, fr)
        hin = [pin32.ptrs[i % 16] for i in range(64)]
        hout = mid.PinnedFrames(ctx, 64, W * H * 16)
        ws = []
        for i in range(6):
            t0 = time.perf_counter()
            ctx.sequence_nlm_pinned(hin, hout.ptrs, W, H, mid.FMT_RGBA32F, k=0, overlap=True, **mid.NLM_BENCH)
            if i >= 2:
                ws.append((time.perf_counter() - t0) * 1e3)
        ws.sort()
        res["f32_frames_64"] = {"Mpixel/s_median": round(64 * W * H / 1e3 / ws[len(ws) // 2], 1), "wall_ms": [round(x, 2) for x in ws]}
        hout.free(); pin32.free()
    print("RESULT " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child")
    ap.add_argument("--timeline")
    ap.add_argument("--only", help="comma-separated variant names")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r6_pipe_u8_ab"))
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.timeline)
    os.makedirs(a.out, exist_ok=True)
    names = a.only.split(",") if a.only else [v[0] for v in VARIANTS]
    results = []
    for rnd in range(a.rounds):
        for name, env in VARIANTS:
            if name not in names:
                continue
            e = dict(os.environ)
            e.pop("GPU_MAX_HW_QUEUES", None); e.pop("MID_PIPE_EXP", None)
            e.update(env)
            tl = os.path.join(a.out, f"timeline_{name}_{rnd}.txt")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--timeline", tl], env=e,
                               capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if p.returncode or not line:
                print(f"== {name} round {rnd}: FAILED rc={p.returncode}\n{p.stdout[-600:]}\n{p.stderr[-1200:]}", flush=True)
                continue
            r = json.loads(line[0][7:])
            results.append(r)
            if "f32_frames_64" in r:
                print(f"== {name:22s} r{rnd} f32 64 frames: {r['f32_frames_64']}", flush=True)
            for nfr in (16, 64):
                if f"frames_{nfr}" not in r:
                    continue
                x = r[f"frames_{nfr}"]
                t = x["timeline"]
                print(f"== {name:22s} r{rnd} {nfr:2d} frames: median {x['Mpixel/s_median']:7.1f} Mpx/s (min/max {x['Mpixel/s_min_max']}, spread {x['spread_pct']} %) "
                      f"kernel avg {t['avg_kernel_ms']:.3f} ms busy {t['kernel_stream_busy_frac']:.2f} | up {t['avg_upload_ms']:.3f} down {t['avg_download_ms']:.3f} ms | "
                      f"gated {t['launches_gated_by']} idle {t['kernel_stream_idle_ms_by_gate']} | sha {x['sha256_of_outputs']}", flush=True)
    json.dump(results, open(os.path.join(a.out, "results.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
