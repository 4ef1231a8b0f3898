#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/ab_libs.sh 3 tools/_build/libdalle_hip_sortA.so tools/_build/libdalle_hip_sortB.so tools/_build/libdalle_hip_sortC.so > gpurun_out/r04y_ab_sort.log 2>&1; cat gpurun_out/r04y_ab_sort.log
