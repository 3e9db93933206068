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
            # the bench launches only: since round 4 the same run also times one-frame launches (config.single_frame_launch), which
            # have a smaller grid -- the timed loop's launches are the ones with the LARGEST grid of the dominant kernel
            cand = [r for r in csv.DictReader(open(tr[0])) if "nlm_strip_kernel" in r["Kernel_Name"]]
            gmax = max(int(r["Grid_Size_X"]) for r in cand) if cand else 0
            d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), short(r["Kernel_Name"]))
                        for r in cand if int(r["Grid_Size_X"]) == gmax))
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

# ---- fingerprint of the profiled bench kernel (tools/run_profiles.sh writes it ON THE GPU BOX from the library it profiled) ----
FP = None
fp_path = os.path.join(src, "code_fingerprint.json")
if os.path.exists(fp_path):
    FP = json.load(open(fp_path))
else:
    print("WARNING: no code_fingerprint.json next to the counters: bench.py will refuse to read these figures back")

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
             **({"kernel_symbol": FP["kernel_symbol"], "kernel_code_sha256": FP["kernel_code_sha256"]} if FP else {}),
             "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"}
        json.dump(t, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
        print(json.dumps(t, indent=1))
# ---- traffic of the bilateral r=8 linear launch (`bench.py --workload bilateral`; north_star: "achieved HBM GB/s (bilateral)") ----
bil = [k for k in pmc if "bilateral_kernel<8, 2, 8, 0, true, 0, mid::BilOne>" in k[0] and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]]
if bil:
    d = pmc[bil[0]]
    fetch_kb = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
    write_kb = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    fpb = (FP or {}).get("workloads", {}).get("bilateral")
    tb = {"kernel": bil[0][0], "grid": bil[0][1], "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
          "read_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
          "traffic_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
          "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
          "algorithmic_bytes_per_launch": 1920 * 1080 * 32,
          "date": __import__("time").strftime("%Y-%m-%d", __import__("time").gmtime()),
          **({"kernel_symbol": fpb["kernel_symbol"], "kernel_code_sha256": fpb["kernel_code_sha256"]} if fpb else {}),
          "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 tools/profile_kernels.py 3 (one 1080p frame per launch, r = 8, linear addressing)"}
    json.dump(tb, open(os.path.join(dst, f"{tag}_traffic_bilateral.json"), "w"), indent=1)
    print(json.dumps(tb, indent=1))
# ---- utilisation from counters (north_star: "LDS/VALU utilisation (NLM, compute-bound) against gfx950 peaks") ------------
# Units (MI355X_MICROARCH.md, rocprofv3 PMC): GRBM_GUI_ACTIVE is summed over the 8 XCDs -> /8 = shader cycles of the launch;
# SQ_BUSY_CYCLES is summed over the 32 shader engines, in cycles; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are per-wave
# sums in QUAD-cycles (x4 = cycles); SQ_INSTS_* count wave-instructions; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are LDS-array
# cycles summed over the 256 CUs.
N_SIMD, N_CU, N_XCD = 1024, 256, 8
# instruction classes of the inner loops, from the ISA (llvm -S census): share of the VALU
# wave-instructions that are DPP adds and transcendentals (v_exp_f32); the rest are plain full-rate fp32/int VALU
CLASSES = {"nlm": {"dpp": 48 / 198, "trans": 8 / 198}, "bilateral": {"dpp": 0.0, "trans": 34 / 420}}
# Issue cost per wave-instruction on one SIMD, cycles.
# "floor": what the hardware can do at best, from tools/microbench17.hip (two DIFFERENT streams on the two waves of a SIMD,
# profiles/r04_microbench17_two_streams.txt): a wave64 VALU instruction occupies the SIMD-32 for 2 cycles; a DPP add costs 4 beside
# another wave's DPP adds but only what a plain instruction costs beside another wave's plain instructions (the arrangement
# the NLM loop's issue priorities set up), so its floor is 2 as well; v_exp_f32 holds the pipe for 8 cycles whatever runs beside it
# (a transcendental stream starves its partner).  Round 3 priced DPP at 4 here -- a same-stream price, not a floor -- and the NLM
# kernels read 1.03: above one.
# "occupancy": the prices the same probe measures at the kernel's own occupancy -- NLM, 2 waves per SIMD: plain instructions in
# 32-bit encodings (v_sub/add/mul/fmac_f32_e32: all of the loop's plain instructions) 2.10, DPP beside plain 2.39, v_exp_f32 8.09
# (one wave can only issue every 4.2-4.9 cycles, so two waves do not reach 2.0; v_fma_f32 in its 64-bit encoding would be 2.39); bilateral, 4-8 waves per SIMD: 2.22 / 8.1 (tools/microbench8.hip).  This is the rate a perfectly fed vector pipe
# would reach with THIS instruction mix at THIS occupancy; the gap to 1 is what waits, tile fills, prologues and tails cost.
COST = {"floor": {"plain": 2.0, "dpp": 2.0, "trans": 8.0},
        "nlm_occupancy": {"plain": 2.10, "dpp": 2.39, "trans": 8.09},     # (the loop's FMAs are v_fmac_f32_e32: the 32-bit-encoding class, ISA census)
        "bilateral_occupancy": {"plain": 2.22, "dpp": 2.22, "trans": 8.1},
        "same_stream": {"plain": 2.0, "dpp": 4.0, "trans": 8.0}}
# search offsets per (pixel, neighbour frame) of the NLM kernels that have a fixed count per wave: cycles per wave-offset below
OFFSETS = {"nlm_bench": 441, "nlm_reference_windows": 196, "nlm_temporal_k2": 441 * 5}   # search offsets a wave walks (temporal: x 5 frames per window)


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


def utilisation(key, cls, offsets=None):
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
        "valu_issue_util_def": "SQ_INSTS_VALU priced per class at the hardware FLOOR (plain 2, DPP add 2 -- beside the other wave's plain "
                               "instructions --, v_exp_f32 8 cycles per wave-instruction; class shares from the ISA census; floor measured by "
                               "tools/microbench17.hip) / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs): the share of SIMD issue cycles the instruction mix needs "
                               "at the very least.  A bound: it cannot exceed 1",
        "valu_issue_util_at_occupancy": round(priced(COST[cls + "_occupancy"]) / simd_cyc, 4),
        "valu_issue_util_at_occupancy_def": "the same priced at what tools/microbench17.hip / microbench8.hip measure at this kernel's occupancy (NLM, 2 waves "
                                            "per SIMD: plain (32-bit encodings, incl. v_fmac_f32) 2.10, DPP beside plain 2.39, v_exp_f32 8.09; bilateral, 4-8 "
                                            "waves: 2.22 / 8.1): the fraction of what a perfectly fed vector pipe would do with this mix at this occupancy",
        "valu_issue_util_at_same_stream_prices": round(priced(COST["same_stream"]) / simd_cyc, 4),
        "valu_issue_util_at_same_stream_prices_def": "round 3's table (DPP add 4): the price of a DPP add when BOTH waves of the SIMD issue DPP adds; not a floor "
                                                     "-- the NLM kernels exceed 1 against it because their DPP adds run beside the partner's plain instructions",
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
    if offsets and "SQ_WAVES" in d:
        per = simd_cyc / (d["SQ_WAVES"] * offsets)              # SIMD cycles per (wave, search offset): both waves of a SIMD counted
        sh_ = CLASSES[cls]
        n_inst = d["SQ_INSTS_VALU"] / (d["SQ_WAVES"] * offsets)  # VALU wave-instructions per wave-offset (198-199 in the hot loop + prologue share)
        floor = n_inst * ((1 - sh_["dpp"] - sh_["trans"]) * 2.0 + sh_["dpp"] * 2.0 + sh_["trans"] * 8.0)
        occ = n_inst * ((1 - sh_["dpp"] - sh_["trans"]) * COST["nlm_occupancy"]["plain"] + sh_["dpp"] * 2.39 + sh_["trans"] * 8.09)
        u["cycles_per_wave_offset"] = {"measured": round(per, 1), "floor": round(floor, 1), "at_occupancy_prices": round(occ, 1),
                                       "gap_to_floor": round(per - floor, 1), "gap_to_occupancy_prices": round(per - occ, 1),
                                       "valu_instructions": round(n_inst, 1),
                                       "def": "SIMD cycles of the launch / (SQ_WAVES x search offsets); floor and occupancy prices as in valu_issue_util*"}
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
        u = utilisation(key, cls, OFFSETS.get(name))
        if u:
            util["kernels"][name] = u
if FP:
    util["bench_kernel_symbol"], util["bench_kernel_code_sha256"] = FP["kernel_symbol"], FP["kernel_code_sha256"]
    util["bench_kernel_code_sha256_by_workload"] = {w_: f_["kernel_code_sha256"] for w_, f_ in FP.get("workloads", {}).items()}
if util["kernels"]:
    json.dump(util, open(os.path.join(dst, f"{tag}_utilisation.json"), "w"), indent=1)
    for n_, u in util["kernels"].items():
        print(n_, {k_: v_ for k_, v_ in u.items() if not k_.endswith("_def") and k_ not in ("kernel",)})
print("wrote", sorted(os.listdir(dst)))
