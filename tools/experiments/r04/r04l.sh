#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
NEW=dalle-mtf_amd/dalle_hip/libdalle_hip.so; O2=tools/_build/libdalle_hip_orig2.so; O28=tools/_build/libdalle_hip_orig28.so
bash tools/ab_libs.sh 3 $O28 $O2 $NEW > gpurun_out/r04l_ab_libs.log 2>&1; cat gpurun_out/r04l_ab_libs.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=nt8p=0,cstream=0" "DALLE_HIP_OPTIONS=nt8p=1,cstream=0" 2 > gpurun_out/r04l_ab_env.log 2>&1
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=nt8p=1,cstream=1" "DALLE_HIP_OPTIONS=nt8p=0,cstream=1" 2 >> gpurun_out/r04l_ab_env.log 2>&1; cat gpurun_out/r04l_ab_env.log
