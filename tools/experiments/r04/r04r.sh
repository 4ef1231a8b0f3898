#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "nt8p or ntr" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 3
timeout 300 python tools/kbench.py k512 2>/dev/null | grep "ffn2-dgrad" > gpurun_out/r04r_kbench_ffn2dgrad.log; cat gpurun_out/r04r_kbench_ffn2dgrad.log
