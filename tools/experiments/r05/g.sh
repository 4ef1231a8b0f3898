#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=1200 -k "nt_ln" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -3
tools/ab_env.sh "DALLE_FUSE_LN=0" "DALLE_FUSE_LN=1" 2 2>&1 | tee gpurun_out/r05g_ab_fuse_ln.log
