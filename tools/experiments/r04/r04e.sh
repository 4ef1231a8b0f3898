#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python tools/phases.py 50816 512 softmax8p 2>&1 | grep -v amdgpu > gpurun_out/r04e_phases_nt8p.log
python tools/phases.py 2048 512 bias8p 2>&1 | grep -v amdgpu >> gpurun_out/r04e_phases_nt8p.log
python tools/phases.py 50816 512 bias8p 2>&1 | grep -v amdgpu >> gpurun_out/r04e_phases_nt8p.log
cat gpurun_out/r04e_phases_nt8p.log
python -m pytest tests/test_headline_parity_gpu.py -q --timeout=900 -k "dalle_example_shape_step" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 4
