#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> per-kernel step breakdown: tools/prof_step.sh <tag> [env assignments...]
# (BENCH_ARGS="--model vae_coco" profiles another model)
TAG=$1; shift
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $OUT.log 2>&1)
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/trace_summary.py $f 3 > /root/repo/gpurun_out/${TAG}_step_breakdown.txt
cp $(find $OUT -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $OUT
head -${PROF_LINES:-30} /root/repo/gpurun_out/${TAG}_step_breakdown.txt
