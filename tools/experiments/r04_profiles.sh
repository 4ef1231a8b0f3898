#!/bin/bash
# Round-4 evidence in one gpurun call: PMC traffic passes first (so that bench.py can attach them), then the bench lines, the
# rocprofv3 kernel summaries (DALL-E step, vae_coco step), per-kernel micro-benchmarks and the GPU suite.  Everything lands under
# gpurun_out/; the author copies it to profiles/.
cd /root/repo; mkdir -p gpurun_out
tools/pmc_traffic.sh r04 "gemm_nt8p_kernel<65>" r04_traffic_vocab_gemm.json 0 "" > gpurun_out/r04_pmc_vocab.log 2>&1
tools/pmc_traffic.sh r04v "conv_gemm_nt_kernel<3>" r04_traffic_vae_coco_conv.json 139198464 524288 --model vae_coco > gpurun_out/r04_pmc_conv.log 2>&1
cp gpurun_out/r04_traffic_vocab_gemm.json gpurun_out/r04_traffic_vae_coco_conv.json profiles/ 2>/dev/null
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err
PROF_LINES=45 tools/prof_step.sh r04 > gpurun_out/r04_prof.log 2>&1
BENCH_ARGS="--model vae_coco" PROF_LINES=45 tools/prof_step.sh r04_vae_coco > gpurun_out/r04_prof_vae_coco.log 2>&1
BENCH_ARGS="--model vae_example" PROF_LINES=30 tools/prof_step.sh r04_vae_example > gpurun_out/r04_prof_vae_example.log 2>&1
for m in vae_example vae_coco; do python bench.py --model $m --steps 100 --warmup 10 > gpurun_out/r04_bench_$m.json 2>/dev/null; done
python bench.py --model 1.3B --steps 20 --warmup 5 > gpurun_out/r04_bench_1p3B.json 2>/dev/null
python bench.py --model dalle_coco --steps 40 --warmup 5 > gpurun_out/r04_bench_dalle_coco.json 2>/dev/null
python tools/kbench.py attn head nt tn k512 n512 2>/dev/null | grep -v amdgpu > gpurun_out/r04_kbench_all.log
tools/pmc_kernel.sh r04_head_p1 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" pmchead > /dev/null 2>&1
tools/pmc_kernel.sh r04_head_p2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE" pmchead > /dev/null 2>&1
for i in 1 2; do python -m pytest tests -m gpu -x -q --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3; done > gpurun_out/r04_pytest_x2.log
tail -n 3 gpurun_out/r04_pytest_x2.log; head -c 900 gpurun_out/r04_bench_n1.json; echo; head -14 gpurun_out/r04_step_breakdown.txt; head -12 gpurun_out/r04_vae_coco_step_breakdown.txt; cat gpurun_out/r04_traffic_vocab_gemm.json
