#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/ab_env.sh "DALLE_SORT_AT_FORWARD=1" "DALLE_SORT_AT_FORWARD=0" 3 > gpurun_out/r04w_ab_sort.log 2>&1; cat gpurun_out/r04w_ab_sort.log
bash tools/r04v.sh 2>&1 | grep -v "^\[gpurun\]" | head -7
python -m pytest tests/test_dalle_step_gpu.py tests/test_headline_parity_gpu.py tests/test_bench_dp_gpu.py -q -x --timeout=900 2>&1 | tail -2
