"""video-learning-tf_amd: the LRCN hot path of npit/video-learning-tf on MI355X (gfx950).

Python host code (mirroring the reference's run_task.py / config.yml / TFRecord surface) over
hand-written HIP kernels reached through the C-ABI of ``libvltf_hip.so`` (include/vltf.h).
Import as ``import vltf_amd`` (see vltf_amd.py at the repo root)."""
__version__ = "0.1.0"
