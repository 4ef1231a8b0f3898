#!/bin/bash
# round-4 call A: GPU suite on the cleanup / oracle changes (no -x: provisional tolerances), baseline bench, NT block timelines
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=1200 2>&1 | grep -v "version\|Hostname\|Librccl" > gpurun_out/r04a_pytest.log
tail -n 25 gpurun_out/r04a_pytest.log
python bench.py > gpurun_out/r04a_bench_n1.json 2> gpurun_out/r04a_bench_n1.err; head -c 700 gpurun_out/r04a_bench_n1.json; echo
python tools/phases.py 50816 512 softmax > gpurun_out/r04a_phases_head_softmax.log 2>&1
python tools/phases.py 50816 512 bias > gpurun_out/r04a_phases_head_bias.log 2>&1
python tools/phases.py 1536 512 bias > gpurun_out/r04a_phases_qkv.log 2>&1
python tools/phases.py 2048 512 bias > gpurun_out/r04a_phases_ffn1.log 2>&1
grep -h "^---\|per k-step\|epilogue\|store drain\|block life\|prologue\|co-resident" gpurun_out/r04a_phases_head_softmax.log gpurun_out/r04a_phases_qkv.log
