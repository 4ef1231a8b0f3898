#!/bin/bash
# round-4 call C: register epilogue with 64-B-contiguous stores (product lib) vs LDS epilogue (epi0)
cd /root/repo; mkdir -p gpurun_out
NEW=dalle-mtf_amd/dalle_hip/libdalle_hip.so; OLD=tools/_build/libdalle_hip_epi0.so
python -m pytest tests/test_kernels_gpu.py -q -x --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 6 > gpurun_out/r04c_pytest_gemm.log
cat gpurun_out/r04c_pytest_gemm.log
(for L in $OLD $NEW $OLD $NEW; do echo "## $L"; DALLE_HIP_LIB=$(realpath $L) python tools/kbench.py k512 2>/dev/null | grep -v amdgpu; done) > gpurun_out/r04c_kbench_k512.log
cat gpurun_out/r04c_kbench_k512.log
bash tools/ab_libs.sh 3 $OLD $NEW > gpurun_out/r04c_ab_step.log 2>&1; cat gpurun_out/r04c_ab_step.log
python tools/phases.py 50816 512 softmax > gpurun_out/r04c_phases_head_softmax.log 2>&1
python tools/phases.py 1536 512 bias > gpurun_out/r04c_phases_qkv.log 2>&1
grep -h "^---\|per k-step\|epilogue\|store drain\|block life\|prologue" gpurun_out/r04c_phases_head_softmax.log gpurun_out/r04c_phases_qkv.log
python -m pytest tests/test_headline_parity_gpu.py tests/test_vae_coco_parity_gpu.py -q --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -n 12 > gpurun_out/r04c_pytest_parity.log; cat gpurun_out/r04c_pytest_parity.log
