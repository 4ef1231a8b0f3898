#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "gemm or softmax_head" 2>&1 | tail -2
(for o in "ntload=0,cstream=4" "ntload=1,cstream=1" "ntload=0,cstream=4" "ntload=1,cstream=1"; do echo "## $o"; KB_OPTIONS=$o timeout 300 python tools/kbench.py head 2>/dev/null | grep "softmax (no\|wgrad\|dgrad"; done) > gpurun_out/r04ad_kbench_nt.log; cat gpurun_out/r04ad_kbench_nt.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=ntload=0,cstream=4" "DALLE_HIP_OPTIONS=ntload=0,cstream=1" 2 2>&1 | tee -a gpurun_out/r04ad_kbench_nt.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=ntload=0,cstream=1" "DALLE_HIP_OPTIONS=ntload=1,cstream=1" 2 2>&1 | tee -a gpurun_out/r04ad_kbench_nt.log
