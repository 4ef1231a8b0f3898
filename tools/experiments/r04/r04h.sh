#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u | tr '\n' ' ') > gpurun_out/r04h_counters.txt
wc -w gpurun_out/r04h_counters.txt
bash tools/pmc_kernel.sh r04h_p1 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" pmchead | grep -A10 "gemm_nt" | head -60
bash tools/pmc_kernel.sh r04h_p2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM" pmchead | grep -A10 "gemm_nt" | head -60
bash tools/pmc_kernel.sh r04h_p3 "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" pmchead | grep -A6 "gemm_nt" | head -40
