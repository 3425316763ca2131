#!/bin/bash
# Builds libvltf_hip.so (gfx950 only) in-tree next to the sources' parent package.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libvltf_hip.so"
obj="$here/obj"
if [ "${VL_EXPERIMENTS:-0}" = 1 ]; then   # the A/B switches of common.h:vl_exp_env compiled in (tools/ only)
  out="$here/../libvltf_hip_exp.so"; obj="$here/obj_exp"; EXP="-DVL_EXPERIMENTS"
fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize: no compiler-formed packed fp32 VALU (v_pk_{add,mul,fma}_f32).  On MI355X the form with op_sel[1] set (low result from
# src1's HIGH register) returns src0.lo + 0 in lanes 48..63 while a split-bf16 kernel (v_cvt_pk / v_pk_add + bf16 MFMA) shares the CU:
# DESIGN 6, tools/ubench/pk_opsel_raw.hip.  tests/test_isa_lint.py checks the built library for that form.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm -fno-slp-vectorize ${EXP:-}"
mkdir -p "$obj"
pids=()
for f in api mfma_gemm pointwise lstm_cluster resize conv_c8; do
  if [ ! -f "$obj/$f.o" ] || [ "$here/$f.hip" -nt "$obj/$f.o" ] || [ "$here/common.h" -nt "$obj/$f.o" ] || [ "$here/conv_desc.h" -nt "$obj/$f.o" ] \
     || [ "$here/../../include/vltf.h" -nt "$obj/$f.o" ] || [ "$here/build.sh" -nt "$obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj/api.o" "$obj/mfma_gemm.o" "$obj/pointwise.o" "$obj/lstm_cluster.o" "$obj/resize.o" "$obj/conv_c8.o"
[ -n "${EXP:-}" ] || gcc -O3 -msse4.2 -std=c11 -fPIC -shared -Wall -pthread -o "$here/../libvltf_host.so" "$here/host_io.c"
echo "built $out and $here/../libvltf_host.so"
