#!/bin/bash
# Round-6 evidence in one gpurun call: PMC traffic of the vocabulary projection first (bench.py attaches it), the bench lines, the
# rocprofv3 kernel summaries, the micro-benchmarks, the attention kernels' PMC counters (last collected in round 2), secondary models.
# (The GPU suite runs in its own call: profiles/r06_pytest_gpu*.log.)
cd /root/repo; mkdir -p gpurun_out
R=${ROUND_TAG:-r06}
tools/pmc_traffic.sh $R "gemm_nt8p_kernel<65>" ${R}_traffic_vocab_gemm.json 0 "" > gpurun_out/${R}_pmc_vocab.log 2>&1
cp gpurun_out/${R}_traffic_vocab_gemm.json profiles/ 2>/dev/null
python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench_n1.err
PROF_LINES=45 tools/prof_step.sh $R > gpurun_out/${R}_prof.log 2>&1
BENCH_ARGS="--model vae_coco" PROF_LINES=45 tools/prof_step.sh ${R}_vae_coco > gpurun_out/${R}_prof_vae_coco.log 2>&1
for m in vae_example vae_coco; do python bench.py --model $m --steps 100 --warmup 10 > gpurun_out/${R}_bench_$m.json 2>/dev/null; done
python bench.py --model 1.3B --steps 20 --warmup 5 > gpurun_out/${R}_bench_1p3B.json 2>/dev/null
python bench.py --model dalle_coco --steps 40 --warmup 5 > gpurun_out/${R}_bench_dalle_coco.json 2>/dev/null
python tools/kbench.py attn head nt tn k512 n512 2>/dev/null | grep -v amdgpu > gpurun_out/${R}_kbench_all.log
tools/pmc_kernel.sh ${R}_attn_p1 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" attn > /dev/null 2>&1
tools/pmc_kernel.sh ${R}_attn_p2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE" attn > /dev/null 2>&1
cat gpurun_out/${R}_attn_p1_pmc.txt gpurun_out/${R}_attn_p2_pmc.txt > gpurun_out/${R}_attn_pmc.txt 2>/dev/null
tools/pmc_kernel.sh ${R}_head_p1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" head > /dev/null 2>&1
tools/pmc_kernel.sh ${R}_tn_p1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" tn > /dev/null 2>&1
python tools/experiments/r06_tn_group_wide.py 2>/dev/null | grep -v amdgpu > gpurun_out/${R}_tn_group_wide.log
head -c 1200 gpurun_out/${R}_bench_n1.json; echo; head -40 gpurun_out/${R}_step_breakdown.txt; for m in vae_example vae_coco 1p3B dalle_coco; do python -c "import json;d=json.load(open('gpurun_out/${R}_bench_$m.json'));print('$m', d['ms_per_step'], d['roofline'].get('step_mfma_frac'))"; done; cat gpurun_out/${R}_attn_pmc.txt | head -30
