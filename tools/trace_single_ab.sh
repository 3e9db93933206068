# kernel trace of lone-frame NLM launches (tools/trace_single.py) for the shipped library and for the libraries named in $LIBS (space separated paths)
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r6_single_shipped -- python3 $R/tools/trace_single.py > $R/gpurun_out/r6_single_shipped.log 2>&1 || exit 1
for L in $LIBS; do
  export MID_LIB_PATH=$R/$L
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r6_single_$(basename $L .so) -- python3 $R/tools/trace_single.py > $R/gpurun_out/r6_single_$(basename $L .so).log 2>&1 || exit 1
done
