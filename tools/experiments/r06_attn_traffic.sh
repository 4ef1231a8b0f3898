#!/bin/bash
# [r06] fabric-side traffic (FETCH_SIZE x 2, WRITE_SIZE; separate --pmc passes, inside the step) of the three attention kernels
cd /root/repo; mkdir -p gpurun_out
# algorithmic bytes per launch at (32, 4, 1280): forward reads qkv once (126 MB) and writes o (42 MB) + lse; dQ reads qkv + o + dO (210 MB), writes dq (42 MB);
# dK/dV reads qkv + dO + stats (168 MB), writes dk, dv (84 MB)
tools/pmc_traffic.sh r06fwd "attn_fwd2_kernel" r06_traffic_attn_fwd.json 168427520 "" > gpurun_out/r06_pmc_attn_fwd.log 2>&1
tools/pmc_traffic.sh r06dq "attn_bwd_dq_kernel" r06_traffic_attn_dq.json 252313600 "" > gpurun_out/r06_pmc_attn_dq.log 2>&1
tools/pmc_traffic.sh r06dkv "attn_bwd_dkv_kernel" r06_traffic_attn_dkv.json 252313600 "" > gpurun_out/r06_pmc_attn_dkv.log 2>&1
for k in fwd dq dkv; do echo "== $k"; cat gpurun_out/r06_traffic_attn_$k.json; echo; done
