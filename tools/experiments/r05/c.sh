#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout=1200 -k "tn_group or small_step or microbatched or recompute_grad" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -12
tools/ab_env.sh "DALLE_WGRAD_PAIR=0" "DALLE_WGRAD_PAIR=1" 2 2>&1 | tee gpurun_out/r05c_ab_wgrad_pair.log
DALLE_WGRAD_PAIR=1 PROF_LINES=24 tools/prof_step.sh r05c | tail -24
