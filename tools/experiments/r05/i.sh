#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=1200 -k "lnbwd or small_step or fused_layernorm_forms or production_dispatch" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl\|version" | tail -4
tools/ab_env.sh "DALLE_LNBWD_CHAIN=0" "DALLE_LNBWD_CHAIN=1" 3 2>&1 | tee gpurun_out/r05i_ab_lnbwd_chain.log
