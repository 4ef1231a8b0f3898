#!/bin/bash
# HBM-side traffic of one kernel of the step from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
#   tools/pmc_traffic.sh <tag> <kernel substring> <out json name> <algorithmic bytes or 0> <grid threads or ""> [bench args...]
TAG=$1; KERNEL=$2; OUT=$3; ALGO=${4:-0}; GRID=$5; shift 5
for C in FETCH_SIZE WRITE_SIZE; do
  D=/root/repo/gpurun_out/pmc_${TAG}_$C
  rm -rf $D; mkdir -p $D
  (cd /tmp && export TMPDIR=/tmp DALLE_VAE_GRAPH=0 && rocprofv3 --pmc $C --output-format csv -d $D -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $D.log 2>&1)
  f=$(find $D -name "*counter_collection.csv" | head -1)
  grep -E "Kernel_Name|$KERNEL" $f | head -400 > /root/repo/gpurun_out/${TAG}_pmc_$C.csv
  rm -rf $D
done
python /root/repo/tools/traffic_summary.py /root/repo/gpurun_out/${TAG}_pmc_FETCH_SIZE.csv /root/repo/gpurun_out/${TAG}_pmc_WRITE_SIZE.csv /root/repo/gpurun_out/$OUT "$KERNEL" $ALGO $GRID
