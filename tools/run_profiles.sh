#!/bin/bash
# Collects the round's rocprofv3 evidence for bench.py on the GPU box (run through gpurun from
# the repo root): kernel-trace/stats, then PMC counters in separate passes (never combined with
# a trace domain), written under gpurun_out/prof_<tag>/.  Summaries are copied to profiles/ by
# tools/summarize_profiles.py afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# sha256 of the bench kernel's code in the library that is about to be profiled: bench.py compares it with the library it
# loads before it reads these counters back (image_denoising_filter_amd/_codeobj.py; pure Python, no GPU)
python3 $R/image_denoising_filter_amd/_codeobj.py $R/image_denoising_filter_amd/libmi_denoise.so > $OUT/code_fingerprint.json || { echo "fingerprint failed"; exit 1; }
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
[ "$2" = "trace_main_only" ] || rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
BENCH2="$BENCH --no-extras"
# the timed loop alone (no side measurements): the dominant kernel's average in this table is the figure to hold against roofline.avg_launch_ms
# (this pass is the DEFAULT command -- 20 timed steps after 3 warm-ups, what the driver runs -- so that its average is the steady state)
BENCH_DEFAULT="python3 $R/bench.py --no-cpu-baseline --no-extras"
echo 3 > $OUT/trace_main.warmup
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_main -- $BENCH_DEFAULT > $OUT/trace_main.log 2>&1 || { echo "trace_main failed"; tail -5 $OUT/trace_main.log; exit 1; }
[ "$2" = "trace_main_only" ] && exit 0
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- $BENCH2 > $OUT/pmc_sq1.log 2>&1 || { echo "pmc_sq1 failed"; tail -5 $OUT/pmc_sq1.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH2 > $OUT/pmc_sq2.log 2>&1 || { echo "pmc_sq2 failed"; tail -5 $OUT/pmc_sq2.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH2 > $OUT/pmc_fetch.log 2>&1 || { echo "pmc_fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH2 > $OUT/pmc_write.log 2>&1 || { echo "pmc_write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
# the same counters for the other kernels of the path (bilateral variants, temporal NLM, streaming passes)
MIX="python3 $R/tools/profile_kernels.py 3"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_mix_sq1 -- $MIX > $OUT/pmc_mix_sq1.log 2>&1 || { echo "pmc_mix_sq1 failed"; tail -5 $OUT/pmc_mix_sq1.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mix_sq2 -- $MIX > $OUT/pmc_mix_sq2.log 2>&1 || { echo "pmc_mix_sq2 failed"; tail -5 $OUT/pmc_mix_sq2.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mix -- $MIX > $OUT/trace_mix.log 2>&1 || { echo "trace_mix failed"; tail -5 $OUT/trace_mix.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_mix_fetch -- $MIX > $OUT/pmc_mix_fetch.log 2>&1 || { echo "pmc_mix_fetch failed"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_mix_write -- $MIX > $OUT/pmc_mix_write.log 2>&1 || { echo "pmc_mix_write failed"; exit 1; }
find $OUT -name "*.csv" | head -30
# the frame pipeline with the library's ROCTx ranges: trace domains only (kernel + memory-copy + marker), no counters;
# summarised by `python tools/pipeline_trace.py --summarise-stages gpurun_out/prof_<tag>/pipe_f32 <tag> f32`
PIPE="python3 $R/tools/pipeline_trace.py"
rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --output-format csv -d $OUT/pipe_f32 -- $PIPE f32 > $OUT/pipe_f32.log 2>&1 || { echo "pipe_f32 failed"; tail -5 $OUT/pipe_f32.log; exit 1; }
rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --output-format csv -d $OUT/pipe_u8 -- $PIPE > $OUT/pipe_u8.log 2>&1 || { echo "pipe_u8 failed"; tail -5 $OUT/pipe_u8.log; exit 1; }
find $OUT -name "*marker*" | head
