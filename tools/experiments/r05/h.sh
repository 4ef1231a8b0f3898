#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
DALLE_HIP_OPTIONS=tn_split_dma=1 python -m pytest tests -m gpu -q --timeout=1200 -k "gemm_tn or wgrad or tn_group or small_step" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -3
tools/ab_env.sh "DALLE_HIP_OPTIONS=tn_split_dma=0" "DALLE_HIP_OPTIONS=tn_split_dma=1" 2 2>&1 | tee gpurun_out/r05h_ab_tn_split_dma.log
DALLE_HIP_OPTIONS=tn_split_dma=0 python tools/kbench.py tn 2>/dev/null | grep "tn8=0" | head -12
DALLE_HIP_OPTIONS=tn_split_dma=1 python tools/kbench.py tn 2>/dev/null | grep "tn8=0" | head -12
