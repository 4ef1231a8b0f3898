#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -4
for i in 1 2; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3))"; DALLE_HIP_OPTIONS=reserve_cus=16 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reserve_cus=16 ms/step', round(d['ms_per_step'],3))"; done
