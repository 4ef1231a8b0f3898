#!/bin/bash
# call B: kernel tests of the changed kernels, dispatch identity, A/B of the bit mask
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout=1200 -k "relu_mask_as_bits or prefetch or production_dispatch or nt_ln or nt8p_persistent or full_row or fused_softmax_head or deferred_finish" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -30 > gpurun_out/r05b_pytest.log
tail -15 gpurun_out/r05b_pytest.log
tools/ab_env.sh "DALLE_HIP_OPTIONS=relu_bits=0" "DALLE_HIP_OPTIONS=relu_bits=1" 2 2>&1 | tee gpurun_out/r05b_ab_relu_bits.log
PROF_LINES=14 tools/prof_step.sh r05b | tail -14
