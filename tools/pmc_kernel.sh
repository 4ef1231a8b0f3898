#!/bin/bash
# per-kernel means of a few PMC counters for a kbench section: tools/pmc_kernel.sh <tag> "<counters>" <kbench section> [KB_* env]
TAG=$1; CNT=$2; SEC=$3
D=/root/repo/gpurun_out/pmck_$TAG
rm -rf $D; mkdir -p $D
(cd /tmp && export TMPDIR=/tmp KB_ITERS=3 KB_WARM_MS=5 && rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $D -- python /root/repo/tools/kbench.py $SEC > $D.log 2>&1)
f=$(find $D -name "*counter_collection.csv" | head -1)
python /root/repo/tools/pmc_summary.py $f | grep -v "at::native\|rocclr" > /root/repo/gpurun_out/${TAG}_pmc.txt
rm -rf $D
cat /root/repo/gpurun_out/${TAG}_pmc.txt
