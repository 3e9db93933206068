"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 CSVs of `bench.py`) into small tracked files under
profiles/:  <tag>_kernel_stats.csv (the --kernel-trace --stats table, library kernels + top-5
others), <tag>_pmc.json (per-kernel counter averages per launch) and <tag>_traffic.json (HBM bytes
per launch of the dominant kernel, with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md 'HBM').  Usage: python tools/summarize_profiles.py r01"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
WARMUP = 2          # tools/run_profiles.sh runs bench.py --steps 5 --warmup 2
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name and name.startswith("mid::") else name[:90]


# ---- kernel stats ----------------------------------------------------------------------------
stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    ours = [r for r in rows if "mid::" in r["Name"]]
    others = [r for r in rows if "mid::" not in r["Name"]][:5]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in ours + others:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    # per-dispatch durations of the dominant kernel by grid size (8-frame launches vs 1-frame ones)
    tr = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
    by = defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        if "mid::" in r["Kernel_Name"]:
            by[(short(r["Kernel_Name"]), r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(dst, f"{tag}_kernel_trace_by_grid.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "LDS_Block_Size", "Launches", "AvgNs", "MinNs", "MaxNs"])
        for k, v in sorted(by.items()):
            w.writerow(list(k) + [len(v), round(sum(v) / len(v)), min(v), max(v)])

main = sorted(glob.glob(os.path.join(src, "trace_main", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if main:   # bench.py --no-extras: only the timed loop's launches, so Calls/AverageNs are those of the bench line
    rows = list(csv.DictReader(open(main[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats_timed_loop.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in [r for r in rows if "mid::" in r["Name"]] + [r for r in rows if "mid::" not in r["Name"]][:5]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
        # the TIMED launches only: the run is `--steps 5 --warmup 2`, so the first 2 launches of the dominant kernel (in
        # start order) are warm-ups; the row below averages the other 5 and is the one to hold against
        # roofline.avg_launch_ms of a bench.py run
        tr = sorted(glob.glob(os.path.join(src, "trace_main", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
        if tr:
            d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), short(r["Kernel_Name"]))
                        for r in csv.DictReader(open(tr[0])) if "nlm_strip_kernel" in r["Kernel_Name"]))
            timed = [x[1] for x in d[WARMUP:]]
            if timed:
                mean = sum(timed) / len(timed)
                sd = (sum((t - mean) ** 2 for t in timed) / len(timed)) ** 0.5
                w.writerow([d[0][2] + f"  [timed launches only: first {WARMUP} of {len(d)} dropped]", len(timed), sum(timed),
                            round(mean), "", min(timed), max(timed), round(sd, 1)])

# ---- PMC -------------------------------------------------------------------------------------
pmc = defaultdict(lambda: defaultdict(list))
newest = {}
for path in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    d_ = os.path.dirname(os.path.dirname(path))
    if d_ not in newest or os.path.getmtime(path) > os.path.getmtime(newest[d_]):
        newest[d_] = path
for path in newest.values():
    seen = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if "mid::" not in r["Kernel_Name"]:
            continue
        key = (short(r["Kernel_Name"]), r["Grid_Size"])
        seen[(key, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (key, _disp, cname), v in seen.items():
        pmc[key][cname].append(v)
out = {}
for key, d in pmc.items():
    out[f"{key[0]} grid={key[1]}"] = {c: {"launches": len(v), "avg_per_launch": sum(v) / len(v)} for c, v in sorted(d.items())}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)

# ---- traffic of the dominant kernel (the 8-frame bench launches = the largest grid) ----------
dom = [k for k in pmc if "nlm_strip_kernel" in k[0]]
if dom:
    k = max(dom, key=lambda kk: int(kk[1]))
    d = pmc[k]
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fetch_kb = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        write_kb = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        t = {"kernel": k[0], "grid": k[1], "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
             "read_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
             "traffic_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
             "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 "
                           "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane streaming stores",
             "algorithmic_bytes_per_launch": int(k[1]) // 256 // (34 * 34) * 1920 * 1080 * 32,
             "date": __import__("time").strftime("%Y-%m-%d", __import__("time").gmtime()),
             "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"}
        json.dump(t, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
        print(json.dumps(t, indent=1))
print("wrote", sorted(os.listdir(dst)))
