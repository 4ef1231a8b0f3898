#!/bin/bash
# Final-tree evidence after the sc1+nt store policy: vocabulary-projection PMC traffic, the default bench line (attaches the
# traffic), the rocprofv3 kernel summary of the same command, the head micro-benchmarks, one pass of the GPU suite.
cd /root/repo; mkdir -p gpurun_out
tools/pmc_traffic.sh r04 "gemm_nt8p_kernel<65>" r04_traffic_vocab_gemm.json 0 "" > gpurun_out/r04_pmc_vocab.log 2>&1
cp gpurun_out/r04_traffic_vocab_gemm.json profiles/ 2>/dev/null
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err
PROF_LINES=45 tools/prof_step.sh r04 > gpurun_out/r04_prof.log 2>&1
python tools/kbench.py head 2>/dev/null | grep -v amdgpu > gpurun_out/r04_kbench_head.log
timeout 420 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3 > gpurun_out/r04_pytest_final.log
cat gpurun_out/r04_pytest_final.log; head -c 1200 gpurun_out/r04_bench_n1.json; echo; head -14 gpurun_out/r04_step_breakdown.txt; cat gpurun_out/r04_traffic_vocab_gemm.json; cat gpurun_out/r04_kbench_head.log
