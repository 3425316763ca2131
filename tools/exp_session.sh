set -e
mkdir -p gpurun_out/r02d
for w in 1 2 3 6; do VL_GEMM_WANT=$w python tools/gemm_probe.py 128 30; done > gpurun_out/r02d/gemm128.txt 2>&1
python tools/gemm_probe.py 1024 10 >> gpurun_out/r02d/gemm128.txt 2>&1
for l in conv2 conv3 conv4 conv5; do for k in fwd dgrad; do
  python tools/conv_probe.py $l $k 128 30; VL_CONV_NO_LOAD_PICK=1 python tools/conv_probe.py $l $k 128 30; done; done > gpurun_out/r02d/conv128.txt 2>&1
python bench.py --clips-per-gpu 8 --no-split-math --no-cpu-baseline --steps 20 > gpurun_out/r02d/bench_c8.json
cat gpurun_out/r02d/gemm128.txt gpurun_out/r02d/conv128.txt
