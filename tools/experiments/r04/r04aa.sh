#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
OUT=/root/repo/gpurun_out/prof_gaps
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python /root/repo/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1)
python tools/gaps.py $(find $OUT -name "*kernel_trace.csv" | head -1) | tee gpurun_out/r04_gaps.txt
rm -rf $OUT
