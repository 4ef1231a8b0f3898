#!/bin/bash
# Final tree of round 4: default bench line, the whole GPU suite, the rocprofv3 kernel summary of the bench command.
cd /root/repo; mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err
head -c 400 gpurun_out/r04_bench_n1.json; echo
timeout 300 python -m pytest tests -m gpu -x -q --timeout=600 > gpurun_out/r04_pytest_final_raw.log 2>&1
grep -v "version\|Hostname\|Librccl" gpurun_out/r04_pytest_final_raw.log | tail -3 | tee gpurun_out/r04_pytest_final.log
PROF_LINES=16 tools/prof_step.sh r04 > gpurun_out/r04_prof.log 2>&1; head -16 gpurun_out/r04_step_breakdown.txt
