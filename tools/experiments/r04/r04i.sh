#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "gemm or softmax_head" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 4
(for o in cstream=0 cstream=2 cstream=0 cstream=2; do echo "## $o"; KB_OPTIONS=$o timeout 600 python tools/kbench.py k512 2>/dev/null | grep "head\|ffn1"; done) > gpurun_out/r04i_kbench_cstream.log; cat gpurun_out/r04i_kbench_cstream.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=cstream=0" "DALLE_HIP_OPTIONS=cstream=1" 2 > gpurun_out/r04i_ab_step.log 2>&1; cat gpurun_out/r04i_ab_step.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=cstream=0" "DALLE_HIP_OPTIONS=cstream=1,cstream_min_mb=100" 2 >> gpurun_out/r04i_ab_step.log 2>&1; tail -4 gpurun_out/r04i_ab_step.log
