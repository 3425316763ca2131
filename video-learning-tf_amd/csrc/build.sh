#!/bin/bash
# Builds libvltf_hip.so (gfx950 only) in-tree next to the sources' parent package.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libvltf_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm"
mkdir -p "$here/obj"
pids=()
for f in api mfma_gemm pointwise lstm_cluster resize conv_c8; do
  if [ ! -f "$here/obj/$f.o" ] || [ "$here/$f.hip" -nt "$here/obj/$f.o" ] || [ "$here/common.h" -nt "$here/obj/$f.o" ] || [ "$here/conv_desc.h" -nt "$here/obj/$f.o" ] \
     || [ "$here/../../include/vltf.h" -nt "$here/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$here/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "$here/obj/api.o" "$here/obj/mfma_gemm.o" "$here/obj/pointwise.o" "$here/obj/lstm_cluster.o" "$here/obj/resize.o" "$here/obj/conv_c8.o"
gcc -O3 -msse4.2 -std=c11 -fPIC -shared -Wall -pthread -o "$here/../libvltf_host.so" "$here/host_io.c"
echo "built $out and $here/../libvltf_host.so"
