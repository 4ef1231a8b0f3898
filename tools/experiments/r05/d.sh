#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout=1200 -k "fused_softmax_head" 2>&1 | grep -v "amdgpu\|Hostname\|Librccl" | tail -6
tools/ab_env.sh "DALLE_HIP_OPTIONS=nt8p_pf=0" "DALLE_HIP_OPTIONS=nt8p_pf=1" 2 2>&1 | tee gpurun_out/r05d_ab_nt8p_pf.log
for pf in 0 1; do DALLE_HIP_OPTIONS=nt8p_pf=$pf python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("pf", '$pf', "head launch_ms", d["roofline"]["launch_ms"], "frac", d["roofline"]["frac"])'; done
