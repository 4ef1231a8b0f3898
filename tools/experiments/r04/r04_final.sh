#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3 > gpurun_out/r04_pytest_final.log; cat gpurun_out/r04_pytest_final.log
python __graft_entry__.py --smoke 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), 'vocab', round(d['roofline']['launch_ms'],3), 'frac', round(d['roofline']['frac'],4))"; done | tee gpurun_out/r04_bench_final_x2.log
