set -eo pipefail
mkdir -p gpurun_out/w2
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/w2/test_gpu.log 2>&1 || { tail -30 gpurun_out/w2/test_gpu.log; exit 1; }
tail -3 gpurun_out/w2/test_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/w2/bench.json 2> gpurun_out/w2/bench.err
cat gpurun_out/w2/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/w2/prof -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/w2/prof_bench.json 2> gpurun_out/w2/prof_bench.err
