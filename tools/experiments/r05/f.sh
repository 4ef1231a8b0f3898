#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=1200 -k "full_row or nt8p_persistent or nt_ln" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -4
tools/ab_env.sh "DALLE_HIP_OPTIONS=res16=0" "DALLE_HIP_OPTIONS=res16=1" 2 2>&1 | tee gpurun_out/r05f_ab_res16.log
DALLE_HIP_OPTIONS=res16=0 KB_COLD=1 python tools/kbench.py n512 2>/dev/null | grep "flags=5.*ntr"
DALLE_HIP_OPTIONS=res16=1 KB_COLD=1 python tools/kbench.py n512 2>/dev/null | grep "flags=5.*ntr"
