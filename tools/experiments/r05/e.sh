#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout=1200 -k "lnbwd or small_step or microbatched or recompute_grad or ref_faithful or golden" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -12
tools/ab_env.sh "DALLE_FUSE_LNBWD=0" "DALLE_FUSE_LNBWD=1" 2 2>&1 | tee gpurun_out/r05e_ab_fuse_lnbwd.log
DALLE_FUSE_LNBWD=1 PROF_LINES=24 tools/prof_step.sh r05e | tail -24
