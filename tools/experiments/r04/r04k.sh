#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
PROF_LINES=40 bash tools/prof_step.sh r04k_new
PROF_LINES=14 bash tools/prof_step.sh r04k_nocs DALLE_HIP_OPTIONS=cstream=0
PROF_LINES=14 bash tools/prof_step.sh r04k_nt4 DALLE_HIP_OPTIONS=nt8p=0,cstream=0
