#!/bin/bash
# alternate two builds of the library on the step benchmark inside one gpurun call: tools/ab.sh <A.so> <B.so> [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    DALLE_HIP_LIB=$(realpath $L) python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],3), round(d['value']))"
  done
done
