set -eo pipefail
mkdir -p gpurun_out/w1
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/w1/test_conv.log 2>&1 || { tail -30 gpurun_out/w1/test_conv.log; exit 1; }
tail -3 gpurun_out/w1/test_conv.log
for l in conv1 conv2 conv3 conv4 conv5; do
  timeout -k 10 120 python tools/conv_probe.py $l wgrad 1024 10 | tee -a gpurun_out/w1/probe.log
  VL_WGRAD_1BUF=1 timeout -k 10 120 python tools/conv_probe.py $l wgrad 1024 10 | sed 's/^/OLD /' | tee -a gpurun_out/w1/probe.log
done
