#!/bin/bash
# One GPU-box pass that produces everything profiles/ is built from (run via gpurun from the repo root):
#   tools/profile_round.sh <tag>          e.g. tools/profile_round.sh r01c
# 1. bench.py (default N=1 run)                               -> gpurun_out/<tag>/bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of the same command     -> gpurun_out/<tag>/prof/
# 3. HBM counters, separate --pmc passes (MI355X_MICROARCH.md "HBM"/"rocprofv3 PMC slots"):
#    FETCH_SIZE and WRITE_SIZE cannot share a pass            -> gpurun_out/<tag>/pmc_{fetch,write}/
# 4. bench.py through the torch.distributed launcher, 1 rank  -> gpurun_out/<tag>/bench_torchrun1.json
# tools/summarize_profile.py turns 2. and 3. into the files committed under profiles/.
set -eo pipefail
tag="${1:-r01}"
out="gpurun_out/$tag"
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
tail -c 400 "$out/bench_n1.json"; echo
rocprofv3 --kernel-trace --stats -d "$out/prof" -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    > "$out/prof_bench.json" 2> "$out/prof_bench.err"
echo "kernel-trace pass done"
for c in FETCH_SIZE WRITE_SIZE; do
  d="$out/pmc_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)"
  rocprofv3 --kernel-trace --pmc $c -d "$d" -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-math \
      > "$d.json" 2> "$d.err"
  echo "pmc pass $c done"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_torchrun1.json" 2> "$out/bench_torchrun1.err"
tail -c 300 "$out/bench_torchrun1.json"; echo
# 5. issue/stall counters of the dominant kernel alone (conv3 forward, the bench's frame count)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
    SQ_INSTS_VALU SQ_INSTS_MFMA -d "$out/pmc_sq_conv3fwd" -o runc -- python3 tools/conv_probe.py conv3 fwd 1024 3 \
    > "$out/pmc_sq_conv3fwd.log" 2>&1
# 6. the same counters for the opt-in bf16x3 kernel of the same launch
VL_CONV_MATH=bf16x3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
    SQ_INSTS_VALU SQ_INSTS_MFMA -d "$out/pmc_sq_conv3fwd_bf16x3" -o runc -- python3 tools/conv_probe.py conv3 fwd 1024 3 \
    > "$out/pmc_sq_conv3fwd_bf16x3.log" 2>&1
echo "profile_round $tag done"
