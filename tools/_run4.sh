set -eo pipefail
mkdir -p gpurun_out/w4
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/w4/test_conv.log 2>&1 || { tail -30 gpurun_out/w4/test_conv.log; exit 1; }
tail -3 gpurun_out/w4/test_conv.log
for l in conv1 conv2 conv3 conv4 conv5; do
  timeout -k 10 120 python tools/conv_probe.py $l fwd 1024 10 | tee -a gpurun_out/w4/probe.log
  if [ $l != conv1 ]; then timeout -k 10 120 python tools/conv_probe.py $l dgrad 1024 10 | tee -a gpurun_out/w4/probe.log; fi
done
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/w4/test_gpu.log 2>&1 || { tail -30 gpurun_out/w4/test_gpu.log; exit 1; }
tail -3 gpurun_out/w4/test_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/w4/bench.json 2> gpurun_out/w4/bench.err
cat gpurun_out/w4/bench.json
