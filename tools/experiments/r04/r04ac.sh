#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
(for o in cstream=2 cstream=3 cstream=2 cstream=3; do echo "## $o"; KB_OPTIONS=$o timeout 300 python tools/kbench.py pmchead 2>/dev/null | grep "nt8p"; done) > gpurun_out/r04ac_kbench_nt_store.log; cat gpurun_out/r04ac_kbench_nt_store.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=cstream=2" "DALLE_HIP_OPTIONS=cstream=3" 2 2>&1 | tee -a gpurun_out/r04ac_kbench_nt_store.log
