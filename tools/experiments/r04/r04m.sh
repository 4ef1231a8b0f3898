#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout=600 -k "ntr or nt8p or nt4_tile or nt8_tile" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 6
timeout 600 python tools/kbench.py n512 2>/dev/null | grep -v amdgpu > gpurun_out/r04m_kbench_n512.log; cat gpurun_out/r04m_kbench_n512.log
