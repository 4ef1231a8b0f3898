#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "nt8p or softmax_head" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 8 > gpurun_out/r04f_pytest_gemm.log
cat gpurun_out/r04f_pytest_gemm.log
python tools/phases.py 50816 512 softmax8p 2>&1 | grep -v amdgpu > gpurun_out/r04f_phases_nt8p.log
python tools/phases.py 2048 512 bias8p 2>&1 | grep -v amdgpu >> gpurun_out/r04f_phases_nt8p.log
cat gpurun_out/r04f_phases_nt8p.log
timeout 600 python tools/kbench.py k512 2>/dev/null | grep -v amdgpu > gpurun_out/r04f_kbench_k512.log; cat gpurun_out/r04f_kbench_k512.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=nt8p=0" "DALLE_HIP_OPTIONS=nt8p=1" 3 > gpurun_out/r04f_ab_step.log 2>&1; cat gpurun_out/r04f_ab_step.log
