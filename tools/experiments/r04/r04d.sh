#!/bin/bash
# round-4 call D: persistent 256x256 NT kernel (option nt8p): correctness, per-kernel, step A/B (auto mode vs off)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "nt8p or softmax_head or nt4_tile or nt8_tile" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 8 > gpurun_out/r04d_pytest_gemm.log
cat gpurun_out/r04d_pytest_gemm.log
timeout 600 python tools/kbench.py k512 2>/dev/null | grep -v amdgpu > gpurun_out/r04d_kbench_k512.log; cat gpurun_out/r04d_kbench_k512.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=nt8p=0" "DALLE_HIP_OPTIONS=nt8p=1" 3 > gpurun_out/r04d_ab_step.log 2>&1; cat gpurun_out/r04d_ab_step.log
