#!/bin/bash
# tools/copy_engine_probe.py three times -- HSA_ENABLE_SDMA unset, 0, 1 -- each plainly and under rocprofv3 --kernel-trace --stats
# (to count the runtime's blit kernels); results under gpurun_out/r4_copy_engine/.  Run through gpurun from the repo root.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4_copy_engine
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in default 0 1; do
  if [ "$mode" = default ]; then unset HSA_ENABLE_SDMA; else export HSA_ENABLE_SDMA=$mode; fi
  echo "== HSA_ENABLE_SDMA=$mode"
  timeout -k 10 150 python3 $R/tools/copy_engine_probe.py > $O/plain_$mode.txt 2>&1 || { echo "plain run failed"; tail -3 $O/plain_$mode.txt; continue; }
  grep -E "pinned copies|pipeline" $O/plain_$mode.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -- python3 $R/tools/copy_engine_probe.py > $O/traced_$mode.txt 2>&1 || { echo "traced run failed"; continue; }
  f=$(ls $O/trace_$mode/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "copyBuffer|nlm_strip|fillBuffer" "$f" < /dev/null | cut -c 1-220
done
