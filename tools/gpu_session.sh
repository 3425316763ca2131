#!/bin/bash
# One GPU-box session (run via gpurun from the repo root):  tools/gpu_session.sh <tag> [steps...]
#   tests   pytest -m gpu (whole suite, one process)            -> gpurun_out/<tag>/pytest.log
#   bench   bench.py default N=1                                -> gpurun_out/<tag>/bench_n1.json
#   c8      bench.py --clips-per-gpu 8 (one rank's shard of the 8-GPU strong-scaling job) + rocprofv3 kernel trace of it
#   fpc32   bench.py --fpc 32 (config 5's clip length)
#   bf16probe  tools/bf16_grad_probe.py --gpu: the bf16 path's gradients against the rounding-aware oracle
# Steps are joined with &&-semantics (set -e): after a failed or killed GPU step nothing else starts.
set -eo pipefail
tag="${1:-r02}"; shift || true
steps="${*:-tests bench c8 fpc32}"
out="gpurun_out/$tag"
mkdir -p "$out"
export TMPDIR=/tmp
for s in $steps; do
  case "$s" in
    tests)
      timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=15 > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
      tail -3 "$out/pytest.log" ;;
    tests_all)
      timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 > "$out/pytest.log" 2>&1 || { tail -60 "$out/pytest.log"; }
      tail -3 "$out/pytest.log" ;;
    bench)
      timeout -k 10 600 python bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err" || { tail -20 "$out/bench_n1.err"; exit 1; }
      tail -c 600 "$out/bench_n1.json"; echo ;;
    c8)
      for c in 8 16 32; do
        timeout -k 10 300 python bench.py --clips-per-gpu $c --no-split-math --no-cpu-baseline --steps 20 > "$out/bench_c$c.json" 2> "$out/bench_c$c.err" \
          || { tail -20 "$out/bench_c$c.err"; exit 1; }
        python - "$out/bench_c$c.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("clips", r["config"]["clips_per_gpu"], "ms/step", r["ms_per_step"], "clips/s", r["value"], "conv stack ms", r["roofline"]["conv_stack"]["ms_per_step"])
PY
      done
      cd /tmp
      timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_c8" -o c8 -- python3 "$GRAFT_REPO_ROOT/bench.py" --clips-per-gpu 8 \
          --no-split-math --no-cpu-baseline --steps 20 > "$GRAFT_REPO_ROOT/$out/prof_c8.json" 2> "$GRAFT_REPO_ROOT/$out/prof_c8.err" || { tail -20 "$GRAFT_REPO_ROOT/$out/prof_c8.err"; exit 1; }
      cd "$GRAFT_REPO_ROOT"
      python tools/summarize_profile.py "$out" "$out/c8" --prof prof_c8 || true ;;
    fpc32)
      timeout -k 10 400 python bench.py --fpc 32 --no-split-math --no-cpu-baseline --steps 5 --warmup 2 > "$out/bench_fpc32.json" 2> "$out/bench_fpc32.err" \
          || { tail -20 "$out/bench_fpc32.err"; exit 1; }
      tail -c 300 "$out/bench_fpc32.json"; echo ;;
    bf16probe)
      timeout -k 10 900 python tools/bf16_grad_probe.py --clips ${PROBE_CLIPS:-8} --gpu ${PROBE_ARGS:-} > "$out/bf16_probe_ref.txt" 2> "$out/bf16_probe_ref.err" || { tail -20 "$out/bf16_probe_ref.err"; exit 1; }
      tail -25 "$out/bf16_probe_ref.txt" ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
  echo "== step $s done"
done
