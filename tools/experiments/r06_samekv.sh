for rep in 1 2 3; do
  python tools/experiments/r06_fwd_time.py 1
  DALLE_HIP_LIB=tools/_build/libdalle_hip_samekv.so python tools/experiments/r06_fwd_time.py 1
  DALLE_HIP_LIB=tools/_build/libdalle_hip_nodma.so python tools/experiments/r06_fwd_time.py 1
done
