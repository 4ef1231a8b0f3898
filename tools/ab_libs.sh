#!/bin/bash
# alternate several builds of the library on the step benchmark inside ONE gpurun call: tools/ab_libs.sh <rounds> <lib.so>...
R=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    DALLE_HIP_LIB=$(realpath $L) python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'ms/step', round(d['ms_per_step'],3), 'vocab gemm ms', round(d['roofline']['launch_ms'],3))"
  done
done
