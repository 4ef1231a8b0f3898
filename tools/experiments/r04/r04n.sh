#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=ntr=0" "DALLE_HIP_OPTIONS=ntr=1" 3 > gpurun_out/r04n_ab_ntr.log 2>&1; cat gpurun_out/r04n_ab_ntr.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 4
