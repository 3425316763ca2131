set -eo pipefail
mkdir -p gpurun_out/w11
rm -f gpurun_out/w11/probe.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/w11/test_gpu.log 2>&1 || { tail -40 gpurun_out/w11/test_gpu.log; exit 1; }
tail -1 gpurun_out/w11/test_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/w11/bench.json 2> gpurun_out/w11/bench.err || { tail -20 gpurun_out/w11/bench.err; exit 1; }
python -c "
import json; r=json.load(open('gpurun_out/w11/bench.json')); print(r['value'], r['ms_per_step'], r['check']); print(r['roofline']['conv_stack']['per_launch_ms'])"
