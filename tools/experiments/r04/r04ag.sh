#!/bin/bash
# two-pass attention backward, second run: the dQ kernel without register spills (transposed-fragment offsets by XOR: 213 / 254
# VGPRs, no scratch; before, every scratch reload in the loop waited with vmcnt(0) for the P / dS stores in flight).
cd /root/repo; mkdir -p gpurun_out
export DALLE_TEST_EXPERIMENTAL=1
timeout 120 python -m pytest tests/test_kernels_gpu.py -x -q -k "two_pass" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5 > gpurun_out/r04ag_pytest_two_pass.log
cat gpurun_out/r04ag_pytest_two_pass.log
timeout 60 python tools/kbench.py attn 2>/dev/null | grep -v amdgpu > gpurun_out/r04ag_kbench_attn.log; cat gpurun_out/r04ag_kbench_attn.log
for arm in "DALLE_ATTN_TWO_PASS=0" "DALLE_ATTN_TWO_PASS=1 DALLE_HIP_OPTIONS=attn_pds_pol=0" "DALLE_ATTN_TWO_PASS=1 DALLE_HIP_OPTIONS=attn_pds_pol=1"; do
  echo "[$arm] $(env $arm timeout 100 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step=%.3f loss=%.5f' % (d['ms_per_step'], d['config']['final_loss']))")"
done 2>&1 | tee gpurun_out/r04ag_ab_two_pass.log
