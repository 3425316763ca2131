set -eo pipefail
mkdir -p gpurun_out/w3
timeout -k 10 300 python tools/wgrad_ab.py 257 2>&1 | tee gpurun_out/w3/ab.log
timeout -k 10 300 python tools/wgrad_ab.py 1024 2>&1 | tee -a gpurun_out/w3/ab.log
for m in new old; do
  if [ $m = old ]; then export VL_WGRAD_1BUF=1; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$m', r['check'], r['value'])" | tee -a gpurun_out/w3/ab.log
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$m', r['check'], r['value'])" | tee -a gpurun_out/w3/ab.log
done
