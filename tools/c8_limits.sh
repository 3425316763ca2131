#!/bin/bash
# Kernel experiments on csrc/conv_c8.hip: builds libvltf_hip variants into scratch/c8exp/ -- run HERE (cross-compile):
#   tools/c8_limits.sh e1="-DC8_EXP=1" spread="-DC8_SPREAD=1" ...      (C8_EXP bits: 1 no in-loop fetches, 2 operands read from LDS
#   once, 4 no MFMAs, 8 no epilogue -- results are garbage, timings are the point)
# then on the GPU box:  VLTF_HIP_LIB=$PWD/scratch/c8exp/libvltf_hip_<name>.so python tools/c8_probe.py 1024 10
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$here/video-learning-tf_amd/csrc"
out="$here/scratch/c8exp"; mkdir -p "$out"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm -Wno-unused-variable -Wno-unused-but-set-variable"
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c "$src/conv_c8.hip" -o "$out/conv_c8_$name.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libvltf_hip_$name.so" "$src/obj/api.o" "$src/obj/mfma_gemm.o" "$src/obj/pointwise.o" \
      "$src/obj/lstm_cluster.o" "$src/obj/resize.o" "$out/conv_c8_$name.o" ) &
done
wait
ls -la "$out"/*.so
