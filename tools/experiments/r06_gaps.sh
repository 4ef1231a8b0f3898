OUT=/root/repo/gpurun_out/prof_gaps
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1)
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/gaps.py $f
rm -rf $OUT
