#!/bin/bash
# two-pass attention backward (P / dS written by the dQ pass, streamed by the dK / dV pass): first run on a GPU.
# kernel tests, micro-benchmark per store policy, per-kernel times, same-call step A/B, the headline parity file with the option on.
cd /root/repo; mkdir -p gpurun_out
export DALLE_TEST_EXPERIMENTAL=1
timeout 240 python -m pytest tests/test_kernels_gpu.py -x -q -k "two_pass" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -15 > gpurun_out/r04af_pytest_two_pass.log
cat gpurun_out/r04af_pytest_two_pass.log
timeout 120 python tools/kbench.py attn 2>/dev/null | grep -v amdgpu > gpurun_out/r04af_kbench_attn.log; cat gpurun_out/r04af_kbench_attn.log
for v in 0 1 0 1; do echo "[DALLE_ATTN_TWO_PASS=$v]"; DALLE_ATTN_TWO_PASS=$v timeout 120 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step=%.3f loss=%.5f' % (d['ms_per_step'], d['config']['final_loss']))"; done 2>&1 | tee gpurun_out/r04af_ab_two_pass.log
cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_attn -o attn -- python /root/repo/tools/kbench.py attn > /dev/null 2>&1; cd /root/repo
f=$(ls /tmp/prof_attn/*/*kernel_stats.csv /tmp/prof_attn/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -i "attn" "$f" | cut -c1-200 > gpurun_out/r04af_attn_kernel_stats.csv; cat gpurun_out/r04af_attn_kernel_stats.csv
DALLE_ATTN_TWO_PASS=1 timeout 200 python -m pytest tests/test_headline_parity_gpu.py -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8 | tee gpurun_out/r04af_pytest_parity_two_pass.log
