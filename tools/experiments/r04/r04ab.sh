#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/ab_libs.sh 3 dalle-mtf_amd/dalle_hip/libdalle_hip.so tools/_build/libdalle_hip_gm4.so tools/_build/libdalle_hip_gm16.so > gpurun_out/r04_ab_groupm.log 2>&1; cat gpurun_out/r04_ab_groupm.log
