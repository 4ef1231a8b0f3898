#!/bin/bash
# round-4 call B: register epilogue (EPI_REGS=1, product lib) vs LDS epilogue (tools/_build/libdalle_hip_epi0.so)
cd /root/repo; mkdir -p gpurun_out
NEW=dalle-mtf_amd/dalle_hip/libdalle_hip.so; OLD=tools/_build/libdalle_hip_epi0.so
# 1. correctness first (GEMM kernels vs fp32 matmul, fused head, model parity)
python -m pytest tests/test_kernels_gpu.py tests/test_headline_parity_gpu.py -q -x --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 15 > gpurun_out/r04b_pytest_gemm.log
cat gpurun_out/r04b_pytest_gemm.log
# 2. per-kernel A/B
(for L in $OLD $NEW $OLD $NEW; do echo "## $L"; DALLE_HIP_LIB=$(realpath $L) python tools/kbench.py k512 2>/dev/null | grep -v amdgpu; done) > gpurun_out/r04b_kbench_k512.log
cat gpurun_out/r04b_kbench_k512.log
# 3. step A/B
bash tools/ab_libs.sh 3 $OLD $NEW > gpurun_out/r04b_ab_step.log 2>&1; cat gpurun_out/r04b_ab_step.log
# 4. block timeline with the register epilogue
python tools/phases.py 50816 512 softmax > gpurun_out/r04b_phases_head_softmax.log 2>&1
grep -h "^---\|per k-step\|epilogue\|store drain\|block life\|prologue" gpurun_out/r04b_phases_head_softmax.log | head -8
# 5. whole GPU suite on the new library
python -m pytest tests -m gpu -q --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 8 > gpurun_out/r04b_pytest_all.log; cat gpurun_out/r04b_pytest_all.log
