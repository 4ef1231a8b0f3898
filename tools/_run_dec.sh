cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_fns_gpu.py -x -q -k "decode or sample" 2>&1 | tail -15 > gpurun_out/dec_tests.log
timeout 600 python tools/samplebench.py 32 > gpurun_out/samplebench_b32.log 2>&1
timeout 300 python tools/samplebench.py 4 > gpurun_out/samplebench_b4.log 2>&1
cat gpurun_out/dec_tests.log gpurun_out/samplebench_b32.log gpurun_out/samplebench_b4.log
