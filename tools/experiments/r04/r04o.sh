#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
PROF_LINES=22 bash tools/prof_step.sh r04o_ntr1 DALLE_HIP_OPTIONS=ntr=1
PROF_LINES=22 bash tools/prof_step.sh r04o_ntr0 DALLE_HIP_OPTIONS=ntr=0
