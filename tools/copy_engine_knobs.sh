#!/bin/bash
# The runtime's copy-engine environment knobs, one at a time, on tools/copy_engine_probe.py (plain run + blit-kernel count from a
# kernel trace).  Exploratory: the knobs are undocumented here (names from `strings libamdhip64.so libhsa-runtime64.so`).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4_copy_engine
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in HSA_REV_COPY_DIR=1 HSA_ENABLE_SDMA_RECOMMENDED_ENG=0 HSA_ENABLE_SDMA_GANG=0 GPU_FORCE_BLIT_COPY_SIZE=0 HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=0; do
  export $kv
  echo "== $kv"
  timeout -k 10 150 python3 $R/tools/copy_engine_probe.py > $O/knob_$kv.txt 2>&1 || { echo "run failed"; tail -3 $O/knob_$kv.txt; unset ${kv%%=*}; continue; }
  grep -E "pinned copies|pipeline" $O/knob_$kv.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$kv -- python3 $R/tools/copy_engine_probe.py > $O/traced_$kv.txt 2>&1
  f=$(ls $O/trace_$kv/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "copyBuffer" "$f" < /dev/null | cut -c 1-120
  unset ${kv%%=*}
done
