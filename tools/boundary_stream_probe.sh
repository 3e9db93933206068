mkdir -p gpurun_out
for hq in default 8; do
  echo "== GPU_MAX_HW_QUEUES=$hq" >> gpurun_out/r6_boundary_low2.txt
  if [ $hq = default ]; then timeout -k 10 200 python tools/boundary_stream_probe.py >> gpurun_out/r6_boundary_low2.txt 2>&1 || exit 1
  else GPU_MAX_HW_QUEUES=$hq timeout -k 10 200 python tools/boundary_stream_probe.py >> gpurun_out/r6_boundary_low2.txt 2>&1 || exit 1; fi
done
