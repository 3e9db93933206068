#!/bin/bash
# A/B of -mllvm -enable-post-misched=false (no post-RA scheduling pass) for nlm.hip / nlm_small.hip: bench.py's headline and lone-frame figures,
# libraries alternated on one lease (build them first: python tools/build_alt.py guess_nopostra nlm.hip -mllvm -enable-post-misched=false; ... small_nopostra nlm_small.hip ...).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for round in 1 2 3; do
  for lib in shipped build/abl/libmi_guess_nopostra.so build/abl/libmi_small_nopostra.so; do
    if [ "$lib" = shipped ]; then unset MID_LIB_PATH; else export MID_LIB_PATH=$PWD/$lib; fi
    python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s value %.1f Mpixel/s  ms_per_step %.4f  lone frame %.4f ms' % ('$lib', d['value'], d['ms_per_step'], d['config']['single_frame_launch']['ms']))"
  done
done
