#!/bin/bash
# Round-3 evidence in one gpurun call: PMC traffic passes first (so that bench.py can attach them), then the bench lines, the
# rocprofv3 kernel summary, the DVFS check and the GPU suite (x3).  Everything lands under gpurun_out/; the author copies it to profiles/.
cd /root/repo; mkdir -p gpurun_out
tools/pmc_traffic.sh r03 "gemm_nt4_kernel<65>" r03_traffic_vocab_gemm.json 0 "" > gpurun_out/r03_pmc_vocab.log 2>&1
tools/pmc_traffic.sh r03v "conv_gemm_nt_kernel<3>" r03_traffic_vae_coco_conv.json 139198464 524288 --model vae_coco > gpurun_out/r03_pmc_conv.log 2>&1
cp gpurun_out/r03_traffic_vocab_gemm.json gpurun_out/r03_traffic_vae_coco_conv.json profiles/ 2>/dev/null
python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err
tools/prof_step.sh r03 > gpurun_out/r03_prof.log 2>&1
for m in vae_example vae_coco; do python bench.py --model $m --steps 100 --warmup 10 > gpurun_out/r03_bench_$m.json 2>/dev/null; done
python bench.py --model 1.3B --steps 20 --warmup 5 > gpurun_out/r03_bench_1p3B.json 2>/dev/null
python bench.py --model dalle_coco --steps 40 --warmup 5 > gpurun_out/r03_bench_dalle_coco.json 2>/dev/null
(echo "# random operands"; python tools/kbench.py big 2>/dev/null | grep "big\|logits"; echo "# zero-filled operands (KB_ZERO=1)"; KB_ZERO=1 python tools/kbench.py big 2>/dev/null | grep "big\|logits") > gpurun_out/r03_kbench_dvfs.log
python tools/kbench.py attn head nt tn 2>/dev/null | grep -v amdgpu > gpurun_out/r03_kbench_all.log
for i in 1 2 3; do python -m pytest tests -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3; done > gpurun_out/r03_pytest_x3.log
tail -n 3 gpurun_out/r03_pytest_x3.log; head -c 600 gpurun_out/r03_bench_n1.json; echo; head -12 gpurun_out/r03_step_breakdown.txt; cat gpurun_out/r03_kbench_dvfs.log
