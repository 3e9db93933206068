R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r6_single_new -- python3 $R/tools/trace_single.py > $R/gpurun_out/r6_single_new.log 2>&1 || exit 1
export MID_LIB_PATH=$R/build/abl/libmi_nlm_ship.so
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r6_single_old -- python3 $R/tools/trace_single.py > $R/gpurun_out/r6_single_old.log 2>&1 || exit 1
