#!/bin/bash
# final-tree check: smoke, the whole GPU suite, the default bench line
cd /root/repo; mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -4 > gpurun_out/r05_smoke.log
python -m pytest tests -m gpu -x -q --timeout=1200 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -6 > gpurun_out/r05_pytest_gpu.log
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
cat gpurun_out/r05_smoke.log; tail -3 gpurun_out/r05_pytest_gpu.log; head -c 700 gpurun_out/r05_bench_final.json; echo; python -c "import json;d=json.load(open('gpurun_out/r05_bench_final.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['cpu_baseline'])"
