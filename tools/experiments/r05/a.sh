#!/bin/bash
# (historical: the ntr_prefetch option it toggles was removed after this call measured it neutral)
# round 5, call A: the tests that round 4 left unrun, the prepared switches (A/B), the cold-cache rows, a baseline profile
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s --timeout=1200 -k "deferred or prefetch or digest or production_dispatch or nt_ln or nt8p_persistent or full_row" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -60 > gpurun_out/r05a_pytest.log
tools/ab_env.sh "DALLE_DEFER_LN=0" "DALLE_DEFER_LN=1" 2 > gpurun_out/r05a_ab_deferln.log 2>&1
tools/ab_env.sh "DALLE_HIP_OPTIONS=ntr_prefetch=0" "DALLE_HIP_OPTIONS=ntr_prefetch=1" 2 > gpurun_out/r05a_ab_prefetch.log 2>&1
KB_COLD=1 python tools/kbench.py n512 2>/dev/null | grep -v amdgpu > gpurun_out/r05a_kbench_n512_cold.log
DALLE_HIP_OPTIONS=ntr_prefetch=1 KB_COLD=1 python tools/kbench.py n512 2>/dev/null | grep -v amdgpu > gpurun_out/r05a_kbench_n512_cold_pf.log
PROF_LINES=45 tools/prof_step.sh r05a > gpurun_out/r05a_prof.log 2>&1
tail -25 gpurun_out/r05a_pytest.log; cat gpurun_out/r05a_ab_deferln.log gpurun_out/r05a_ab_prefetch.log; cat gpurun_out/r05a_kbench_n512_cold.log gpurun_out/r05a_kbench_n512_cold_pf.log; head -12 gpurun_out/r05a_step_breakdown.txt
