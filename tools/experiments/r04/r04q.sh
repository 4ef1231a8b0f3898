#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py -q -x --timeout=600 -k "nt_ln or abi or symbol" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 6
timeout 300 python tools/kbench.py ln512 2>/dev/null | grep -v amdgpu > gpurun_out/r04q_kbench_ln512.log; cat gpurun_out/r04q_kbench_ln512.log
timeout 900 python -m pytest tests/test_headline_parity_gpu.py tests/test_engine_gpu.py -q -x --timeout=900 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 5
bash tools/ab_env.sh "DALLE_FUSE_LN=0" "DALLE_FUSE_LN=1" 3 > gpurun_out/r04q_ab_fuse_ln.log 2>&1; cat gpurun_out/r04q_ab_fuse_ln.log
