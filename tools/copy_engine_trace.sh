#!/bin/bash
# One traced run of tools/copy_engine_probe.py with the runtime's defaults: kernel trace + memory-copy trace, to see WHICH of the
# pipeline's copies the runtime hands to a blit kernel (__amd_rocclr_copyBuffer) and which to the SDMA engines.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4_copy_engine
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
unset HSA_ENABLE_SDMA
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_memcpy -- python3 $R/tools/copy_engine_probe.py > $O/traced_memcpy.txt 2>&1 || { echo "traced run failed"; tail -5 $O/traced_memcpy.txt; exit 1; }
ls $O/trace_memcpy/*/
python3 - <<PY
import csv, glob, collections
d = glob.glob("$O/trace_memcpy/*/*_memory_copy_trace.csv")
k = glob.glob("$O/trace_memcpy/*/*_kernel_trace.csv")
if d:
    rows = list(csv.DictReader(open(d[0])))
    print("memory-copy records:", len(rows), "columns:", list(rows[0].keys()) if rows else None)
    c = collections.Counter()
    for r in rows:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c[(r.get("Direction"), r.get("Bytes") or r.get("Size"))] += 1
    for key, n in sorted(c.items(), key=lambda kv: -kv[1])[:20]:
        print("  ", key, n)
if k:
    rows = [r for r in csv.DictReader(open(k[0])) if "copyBuffer" in r["Kernel_Name"]]
    c = collections.Counter((r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows)
    print("copyBuffer kernels:", len(rows), "by grid:", c.most_common(8))
PY
