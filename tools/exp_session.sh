set -e
mkdir -p gpurun_out/r02j
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv" > gpurun_out/r02j/pytest_conv.log 2>&1 || { tail -30 gpurun_out/r02j/pytest_conv.log; exit 1; }
tail -1 gpurun_out/r02j/pytest_conv.log
{
for p in VL_E=1 VL_CONV_NO_WIDE_STORE=1; do
  echo "$p"; env $p python tools/conv_probe.py conv1 fwd 1024 20; env $p python tools/conv_probe.py conv1 fwd 128 20
done
for l in "conv2 fwd" "conv3 fwd" "conv4 fwd" "conv5 fwd" "conv2 dgrad" "conv3 dgrad" "conv4 dgrad" "conv5 dgrad"; do python tools/conv_probe.py $l 1024 20; done
} 2>&1 | grep -v amdgpu.ids
python bench.py --no-split-math --no-cpu-baseline --steps 10 > gpurun_out/r02j/bench.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r02j/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['conv_stack'], r['oracle_check'])
PY
