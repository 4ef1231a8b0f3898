#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/ab_libs.sh 3 tools/_build/libdalle_hip_nomask.so dalle-mtf_amd/dalle_hip/libdalle_hip.so > gpurun_out/r04s_ab_mask.log 2>&1; cat gpurun_out/r04s_ab_mask.log
