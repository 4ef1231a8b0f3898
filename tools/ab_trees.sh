#!/bin/bash
# alternate the step benchmark of two source trees inside one gpurun call: tools/ab_trees.sh <treeA> <treeB> [rounds] [extra bench args]
A=$1; B=$2; R=${3:-2}; shift $(( $# < 3 ? $# : 3 ))
for i in $(seq 1 $R); do
  for T in "$A" "$B"; do
    (cd $T && python bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T', round(d['ms_per_step'],3), 'vocab gemm ms', round(d['roofline']['launch_ms'],3))")
  done
done
