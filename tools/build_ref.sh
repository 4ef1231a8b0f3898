#!/bin/bash
# Build libdalle_hip from the csrc/ of a given git revision into ab/libdalle_hip_<name>.so (A/B timing of two builds in
# one gpurun call: DALLE_HIP_LIB=ab/libdalle_hip_<name>.so python bench.py ...).  usage: tools/build_ref.sh <rev> <name>
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}; name=${2:-old}
root=$(mktemp -d); tmp="$root/a/b"; mkdir -p "$tmp" "$root/include"   # sources include "../../include/dalle_hip.h"
for f in elementwise.hip gemm.hip attention.hip vae.hip common.h; do git show "$rev:dalle-mtf_amd/csrc/$f" > "$tmp/$f"; done
git show "$rev:include/dalle_hip.h" > "$root/include/dalle_hip.h"
mkdir -p ab
objs=""
for f in elementwise gemm attention vae; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -c "$tmp/$f.hip" -o "$tmp/$f.o" &
  objs="$objs $tmp/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "ab/libdalle_hip_$name.so" $objs
rm -rf "$root"
echo "ab/libdalle_hip_$name.so"
