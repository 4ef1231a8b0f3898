#!/bin/bash
# A/B an engine-level environment toggle on the step benchmark: tools/ab_env.sh VAR [rounds]
V=${1:-DALLE_GROUPED_WGRAD}; R=${2:-2}
for i in $(seq 1 $R); do for g in 0 1; do env $V=$g python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$g', round(d['ms_per_step'],3))"; done; done
