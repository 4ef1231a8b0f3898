#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
(for st in 0 2 4 5 8 0 4; do echo "## nt8p_stagger=$st"; KB_OPTIONS=nt8p_stagger=$st timeout 300 python tools/kbench.py pmchead 2>/dev/null | grep "nt8p"; done) > gpurun_out/r04u_kbench_stagger.log; cat gpurun_out/r04u_kbench_stagger.log
for st in 0 4; do DALLE_HIP_OPTIONS=nt8p_stagger=$st python tools/phases.py 50816 512 softmax8p 2>&1 | grep -v amdgpu | head -2; done
