#!/bin/bash
# Round-4 measurement pass (GPU box, from the repo root, via gpurun): everything profiles/r04_* is made from.  Databases are
# summarised on the box and deleted (gpurun copies back at most 64 MiB).   bash tools/round4_measure.sh <part: a|b|c>
set -o pipefail
part="${1:-a}"
out=gpurun_out/r04_final; mkdir -p $out
export TMPDIR=/tmp
J='import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], "ms/step", d["ms_per_step"], "clips/s", d["value"])'
if [ "$part" = a ]; then
  # 1. the driver's command
  python bench.py > $out/bench_n1.json 2> $out/bench_n1.err || { tail -5 $out/bench_n1.err; exit 1; }
  python -c "$J" n1 < $out/bench_n1.json
  # 2. rocprofv3 kernel trace of the same command (two-stream default) -> kernel stats, alone stats, timeline
  rocprofv3 --kernel-trace --stats -d $out/prof -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_bench.json 2> $out/prof_bench.err || { echo prof failed; exit 1; }
  # 3. HBM counters, separate passes
  for c in FETCH_SIZE WRITE_SIZE; do
    d="$out/pmc_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)"
    rocprofv3 --kernel-trace --pmc $c -d "$d" -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-math > "$d.json" 2> "$d.err" || { echo pmc $c failed; exit 1; }
  done
  python tools/summarize_profile.py $out $out/n1 | tail -3
  python tools/timeline.py $out/prof --step 13 > $out/timeline_clips64.txt 2>&1; tail -1 $out/timeline_clips64.txt
  python tools/timeline.py $out/prof --step 5 > $out/timeline_clips64_two_streams.txt 2>&1; tail -1 $out/timeline_clips64_two_streams.txt
  find $out -name "*_results.db" -delete
  # 4. one rank under the launcher, and the self-launch path with one rank
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-split-math > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err
  python -c "$J" torchrun1 < $out/bench_torchrun1.json
fi
if [ "$part" = b ]; then
  for c in 8 16 32; do
    python bench.py --clips-per-gpu $c --no-split-math --no-cpu-baseline --steps 30 --warmup 5 > $out/bench_clips$c.json 2> $out/bench_clips$c.err; python -c "$J" clips$c < $out/bench_clips$c.json
  done
  rocprofv3 --kernel-trace --stats -d $out/prof_c8 -o runc -- python3 bench.py --clips-per-gpu 8 --no-split-math --no-cpu-baseline --steps 10 --warmup 2 > $out/prof_c8.json 2> $out/prof_c8.err
  python tools/summarize_profile.py $out $out/c8 --prof prof_c8 | tail -2
  python tools/timeline.py $out/prof_c8 --step 17 > $out/timeline_clips8.txt 2>&1; tail -1 $out/timeline_clips8.txt
  python tools/timeline.py $out/prof_c8 --step 7 > $out/timeline_clips8_two_streams.txt 2>&1; tail -1 $out/timeline_clips8_two_streams.txt   # a step of the timed region
  find $out -name "*_results.db" -delete
  python bench.py --fpc 32 --no-split-math --no-cpu-baseline --steps 10 > $out/bench_fpc32.json 2> $out/bench_fpc32.err; python -c "$J" fpc32 < $out/bench_fpc32.json
  python bench.py --conv-math bf16 --no-split-math --no-cpu-baseline --steps 20 > $out/bench_bf16_path_fpc16.json 2> $out/bench_bf16_16.err; python -c "$J" bf16_fpc16 < $out/bench_bf16_path_fpc16.json
  python bench.py --conv-math bf16 --fpc 32 --no-split-math --no-cpu-baseline --steps 10 > $out/bench_bf16_path_fpc32.json 2> $out/bench_bf16_32.err; python -c "$J" bf16_fpc32 < $out/bench_bf16_path_fpc32.json
  python bench.py --conv-math bf16x3 --no-split-math --no-cpu-baseline --steps 20 > $out/bench_bf16x3.json 2> $out/bench_bf16x3.err; python -c "$J" bf16x3 < $out/bench_bf16x3.json
  python tools/bench_composed.py 64 10 > $out/bench_config4.json 2> $out/bench_config4.err; tail -c 300 $out/bench_config4.json; echo
  # the 2-rank launch path for real (gloo, both ranks on this GPU): a rehearsal of the code path, not a timing
  VLTF_DIST_BACKEND=gloo VLTF_WGRAD_STREAM=0 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-split-math > $out/bench_gloo2_self_launched.json 2> $out/bench_gloo2.err; python -c "$J" gloo2 < $out/bench_gloo2_self_launched.json
fi
if [ "$part" = c ]; then
  MFMA_OUT=r04_final/mfma bash tools/mfma_busy_probe.sh > $out/mfma_busy_and_clock.txt 2>&1; cat $out/mfma_busy_and_clock.txt | tail -8
  C8_OUT=r04_final/c8pmc bash tools/c8_pmc_probe.sh > $out/bf16_path_conv_pmc.txt 2>&1; tail -14 $out/bf16_path_conv_pmc.txt
  bash tools/pw_pmc_probe.sh r04_final/pwpmc > $out/pw_pmc.log 2>&1; tail -12 $out/pw_pmc.log
  find $out -name "*_results.db" -delete
fi
