set -eo pipefail
mkdir -p gpurun_out/w8
rm -f gpurun_out/w8/probe.log
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pool_lrn" > gpurun_out/w8/test.log 2>&1 || { tail -30 gpurun_out/w8/test.log; exit 1; }
tail -1 gpurun_out/w8/test.log
for l in 1 2; do
  timeout -k 10 120 python tools/pw_probe.py pool_lrn_bwd $l 1024 10 | tee -a gpurun_out/w8/probe.log
  VL_POOL_LRN_CHK16=1 timeout -k 10 120 python tools/pw_probe.py pool_lrn_bwd $l 1024 10 | sed 's/^/CHK16 /' | tee -a gpurun_out/w8/probe.log
  VL_POOL_LRN_CHUNKED=1 timeout -k 10 120 python tools/pw_probe.py pool_lrn_bwd $l 1024 10 | sed 's/^/OLD /' | tee -a gpurun_out/w8/probe.log
done
