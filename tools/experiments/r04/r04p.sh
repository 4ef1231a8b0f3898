#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=ntr=0" "DALLE_HIP_OPTIONS=ntr=1" 3 > gpurun_out/r04p_ab_ntr.log 2>&1; cat gpurun_out/r04p_ab_ntr.log
bash tools/ab_env.sh "DALLE_HIP_OPTIONS=ntr=0,nt8p=0,cstream=0" "DALLE_HIP_OPTIONS=ntr=1" 2 >> gpurun_out/r04p_ab_ntr.log 2>&1; tail -4 gpurun_out/r04p_ab_ntr.log
