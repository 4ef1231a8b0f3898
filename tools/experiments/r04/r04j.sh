#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
NEW=dalle-mtf_amd/dalle_hip/libdalle_hip.so; OLD=tools/_build/libdalle_hip_epi0.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vae_tokens_gpu.py -q -x --timeout=900 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 4
(for L in $OLD $NEW; do echo "## $L"; DALLE_HIP_LIB=$(realpath $L) timeout 900 python tools/kbench.py nt big 2>/dev/null | grep -v amdgpu; done) > gpurun_out/r04j_kbench_nt.log; cat gpurun_out/r04j_kbench_nt.log
bash tools/ab_libs.sh 3 $OLD $NEW > gpurun_out/r04j_ab_step.log 2>&1; cat gpurun_out/r04j_ab_step.log
