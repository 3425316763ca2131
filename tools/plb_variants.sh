#!/bin/bash
# Variant builds of libvltf_hip.so for the pool_lrn_bwd determinism hunt (tools/plb_race_probe.py): only pointwise.hip is rebuilt,
# with an experiment macro or a code-generation flag; the other objects are the in-tree ones.  Output: scratch/plbv/libvltf_hip_<name>.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/video-learning-tf_amd/csrc"
out="$root/scratch/plbv"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm"
build() {   # name, extra flags
  name="$1"; shift
  $HIPCC $FLAGS "$@" -c "$src/pointwise.hip" -o "$out/pointwise_$name.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libvltf_hip_$name.so" "$src/obj/api.o" "$src/obj/mfma_gemm.o" "$out/pointwise_$name.o" \
      "$src/obj/lstm_cluster.o" "$src/obj/resize.o" "$src/obj/conv_c8.o"
  echo "built $out/libvltf_hip_$name.so"
}
for v in "$@"; do
  case "$v" in
    noslp)   build noslp -fno-slp-vectorize & ;;
    lds2k)   build lds2k -DVL_PLB_EXP=1 & ;;
    stwait)  build stwait -DVL_PLB_EXP=2 & ;;
    schedb)  build schedb -DVL_PLB_EXP=4 & ;;
    stnop)   build stnop -DVL_PLB_EXP=8 & ;;
    o1)      build o1 -O1 & ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
wait
