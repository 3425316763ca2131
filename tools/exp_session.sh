set -e
{
for l in fc6x1024 fc6x128 gxx1024; do for k in fwd dgrad wgrad; do python tools/conv_probe.py $l $k 1 20; done; done
} 2>&1 | grep -v amdgpu.ids
