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
        # the TIMED launches only: the first `warmup` launches of the dominant kernel (in start order) are warm-ups; the row
        # below averages the others and is the one to hold against roofline.avg_launch_ms of a bench.py run
        tr = sorted(glob.glob(os.path.join(src, "trace_main", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
        if tr:
            d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), short(r["Kernel_Name"]))
                        for r in csv.DictReader(open(tr[0])) if "nlm_strip_kernel" in r["Kernel_Name"]))
            # (tools/run_profiles.sh records the pass's warm-up count: since round 3 the trace_main pass is the DEFAULT command,
            # --steps 20 --warmup 3, the one the driver times; older collections used --steps 5 --warmup 2)
            wfile = os.path.join(src, "trace_main.warmup")
            WARMUP = int(open(wfile).read().strip()) if os.path.exists(wfile) else 2
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
# ---- utilisation from counters (north_star: "LDS/VALU utilisation (NLM, compute-bound) against gfx950 peaks") ------------
# Units (MI355X_MICROARCH.md, rocprofv3 PMC): GRBM_GUI_ACTIVE is summed over the 8 XCDs -> /8 = shader cycles of the launch;
# SQ_BUSY_CYCLES is summed over the 32 shader engines, in cycles; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are per-wave
# sums in QUAD-cycles (x4 = cycles); SQ_INSTS_* count wave-instructions; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are LDS-array
# cycles summed over the 256 CUs.
N_SIMD, N_CU, N_XCD = 1024, 256, 8
# instruction classes of the inner loops, from the ISA (llvm -S census): share of the VALU
# wave-instructions that are DPP adds and transcendentals (v_exp_f32); the rest are plain full-rate fp32/int VALU
CLASSES = {"nlm": {"dpp": 48 / 198, "trans": 8 / 198}, "bilateral": {"dpp": 0.0, "trans": 34 / 420}}
# issue cost per wave-instruction on one SIMD, cycles.  "floor": the hardware's best case with several waves per SIMD
# (MI355X_MICROARCH.md: v_fma_f32 wave64 2, transcendentals 8; DPP adds are half rate: 4).  "measured": what
# tools/microbench.hip / microbench5.hip measured at this kernel's occupancy: 2.9 / 4.85 / 8.4 at 2 waves per
# SIMD (NLM), 2.5 / - / 8.4 at 8 waves per SIMD (bilateral).
COST = {"floor": {"plain": 2.0, "dpp": 4.0, "trans": 8.0},
        "nlm_measured": {"plain": 2.9, "dpp": 4.85, "trans": 8.4}, "bilateral_measured": {"plain": 2.5, "dpp": 4.4, "trans": 8.4},
        # tools/microbench8.hip (round 3): cycles counted by the chip (s_memtime), clock measured beside them -- plain = the
        # mix-weighted mean of two-source ops (2.25) and three-VGPR-source FMAs (2.54 at 2 waves/SIMD, 2.22 at 8)
        "nlm_cycle_exact": {"plain": (82 * 2.25 + 60 * 2.54) / 142, "dpp": 4.39, "trans": 8.2},
        "bilateral_cycle_exact": {"plain": 2.22, "dpp": 4.1, "trans": 8.1}}


def durations_by_kernel():
    out = defaultdict(list)
    for sub in ("trace_main", "trace_mix", "trace"):
        tr = sorted(glob.glob(os.path.join(src, sub, "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
        if not tr:
            continue
        for r in csv.DictReader(open(tr[0])):
            if "mid::" in r["Kernel_Name"]:
                key = (short(r["Kernel_Name"]), r["Grid_Size_X"])
                if (sub, key) not in out:
                    out[(sub, key)] = []
                out[(sub, key)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    best = {}
    for (sub, key), v in out.items():
        if key not in best:                      # trace_main first (timed loop only), then the mix, then the full bench trace
            v = sorted(v)
            best[key] = (sum(v[: max(1, len(v) * 3 // 4)]) / max(1, len(v) * 3 // 4), sub, len(v))   # mean of the fastest 3/4: drops cold first launches
    return best


def utilisation(key, cls):
    d = {c: sum(v) / len(v) for c, v in pmc[key].items()}
    need = ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE")
    if any(c not in d for c in need):
        return None
    cyc = d["GRBM_GUI_ACTIVE"] / N_XCD                      # shader cycles of one launch
    simd_cyc, cu_cyc = cyc * N_SIMD, cyc * N_CU
    sh = CLASSES[cls]
    def priced(cost):
        return d["SQ_INSTS_VALU"] * ((1 - sh["dpp"] - sh["trans"]) * cost["plain"] + sh["dpp"] * cost["dpp"] + sh["trans"] * cost["trans"])
    dur = durations_by_kernel().get(key)
    u = {
        "kernel": key[0], "grid": key[1],
        "shader_cycles_per_launch": round(cyc),
        "valu_wave_instructions_per_launch": round(d["SQ_INSTS_VALU"]),
        "valu_issue_util": round(priced(COST["floor"]) / simd_cyc, 4),
        "valu_issue_util_def": "SQ_INSTS_VALU priced per class at the hardware's best-case issue cost (plain 2, DPP add 4, v_exp_f32 8 cycles per "
                               "wave-instruction; class shares from the ISA census) / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs): the share of SIMD issue cycles "
                               "the instruction mix would need on an ideally fed vector pipe",
        "valu_issue_util_at_measured_costs": round(priced(COST[cls + "_measured"]) / simd_cyc, 4),
        "valu_issue_util_at_measured_costs_def": "the same priced at the per-instruction issue costs the micro-benchmarks measured at this kernel's "
                                                 "occupancy (tools/microbench*.hip, DESIGN.md 3.1): ~1.0 means the kernel sits at the issue limit of its mix",
        "valu_issue_util_at_cycle_exact_costs": round(priced(COST[cls + "_cycle_exact"]) / simd_cyc, 4),
        "valu_issue_util_at_cycle_exact_costs_def": "the same priced at the issue costs tools/microbench8.hip measures in shader cycles at this kernel's "
                                                    "occupancy (s_memtime; NLM at 2 waves/SIMD: two-source plain 2.25, three-source FMA 2.54, DPP add 4.39, "
                                                    "v_exp_f32 8.2): the fraction of a perfectly fed vector pipe the kernel reaches at its instruction mix",
        "valu_active_share_of_wave_cycles": round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], 4),
        "valu_busy_gfx94x_formula": round(d["SQ_ACTIVE_INST_VALU"] * 4 / simd_cyc, 4),
        "valu_busy_gfx94x_formula_def": "SQ_ACTIVE_INST_VALU (quad-cycles, per wave) x4 / SIMD-cycles, rocprof's derived VALUBusy for gfx94x; it sums over "
                                        "the waves of a SIMD, whose VALU instructions overlap in the pipeline, so it can exceed 1",
        "waves_per_simd_avg": round(d["SQ_WAVE_CYCLES"] * 4 / simd_cyc, 3),
        "lds_util": round(d["SQ_LDS_IDX_ACTIVE"] / cu_cyc, 4),
        "lds_util_def": "SQ_LDS_IDX_ACTIVE (LDS-array cycles, all CUs) / (GRBM_GUI_ACTIVE/8 x 256 CUs)",
        "lds_bank_conflict_cycles": round(d.get("SQ_LDS_BANK_CONFLICT", 0.0)),
        "lds_bank_conflict_share_of_lds_cycles": round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(d["SQ_LDS_IDX_ACTIVE"], 1.0), 5),
    }
    if "SQ_WAIT_INST_ANY" in d:
        u["issue_stall_share_of_wave_cycles"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)       # SQ_WAIT_INST_ANY: waiting for an issue slot / pipe
        u["parked_share_of_wave_cycles"] = round(d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)       # SQ_WAIT_ANY: parked on s_waitcnt / barrier
        u["active_any_share_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
    else:
        u["issue_stall_share_of_wave_cycles"] = None
    if "SQ_BUSY_CYCLES" in d:
        u["valu_active_quadcycles_per_sq_busy_cycle"] = round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_BUSY_CYCLES"], 3)
        u["valu_active_quadcycles_per_sq_busy_cycle_def"] = ("SQ_ACTIVE_INST_VALU [quad-cycles of wave time, summed over waves] / SQ_BUSY_CYCLES [cycles, summed "
                                                              "over 32 shader engines]; x4/32 SIMDs per engine = valu_busy_gfx94x_formula")
    if dur:
        u["avg_launch_us_in_trace"] = round(dur[0] / 1e3, 1)
        u["effective_clock_GHz"] = round(cyc / dur[0], 3)        # cycles / ns
        u["duration_source"] = f"{dur[1]} kernel trace, mean of the fastest 3/4 of {dur[2]} launches (PMC passes and traces are separate runs)"
    return u


def pick(sub, grid=None, largest=False):
    c = [k for k in pmc if sub in k[0] and "GRBM_GUI_ACTIVE" in pmc[k] and (grid is None or k[1] == grid)]
    if not c:
        return None
    return max(c, key=lambda kk: int(kk[1])) if largest else c[0]


util = {"date": __import__("time").strftime("%Y-%m-%d", __import__("time").gmtime()),
        "commands": "tools/run_profiles.sh " + tag + ": rocprofv3 --pmc passes (SQ pass 1, SQ pass 2 + GRBM_GUI_ACTIVE; each its own run, never with a trace "
                    "domain) of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras` (NLM bench kernel) and of tools/profile_kernels.py (the others)",
        "peaks": {"simds": N_SIMD, "cus": N_CU, "note": "utilisations are fractions of SIMD issue cycles / CU LDS cycles of the launch itself (cycles from "
                                                          "GRBM_GUI_ACTIVE, i.e. at the clock the chip actually held)"},
        "kernels": {}}
for name, key, cls in (("nlm_bench", pick("nlm_strip_kernel<-10, 11, -3, 4, 8, 4, 0, true, false", largest=True), "nlm"),
                       ("nlm_temporal_k2", pick("nlm_strip_kernel<-10, 11, -3, 4, 8, 4, 0, true, true"), "nlm"),
                       ("nlm_reference_windows", pick("nlm_strip_kernel<-7, 7, -3, 3, 8, 4, 0, true, false"), "nlm"),
                       ("bilateral_r8_linear", pick("bilateral_kernel<8, 2, 8, 0, true, 0, mid::BilOne"), "bilateral"),
                       ("bilateral_r8_texture", pick("bilateral_kernel<8, 2, 8, 0, false, 0, mid::BilOne"), "bilateral"),
                       ("bilateral_r20_texture", pick("bilateral_kernel<20, 1, 8, 0, false, 0, mid::BilOne"), "bilateral"),
                       ("bilateral_layers_r8_L4_fused", pick("bilateral_kernel<8, 2, 8, 0, false, 2, mid::BilOne"), "bilateral")):
    if key:
        u = utilisation(key, cls)
        if u:
            util["kernels"][name] = u
if util["kernels"]:
    json.dump(util, open(os.path.join(dst, f"{tag}_utilisation.json"), "w"), indent=1)
    for n_, u in util["kernels"].items():
        print(n_, {k_: v_ for k_, v_ in u.items() if not k_.endswith("_def") and k_ not in ("kernel",)})
print("wrote", sorted(os.listdir(dst)))
