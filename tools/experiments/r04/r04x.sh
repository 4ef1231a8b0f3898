#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "sort or embed" 2>&1 | tail -2
bash tools/r04v.sh 2>&1 | head -22
for i in 1 2 3; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3))"; done
