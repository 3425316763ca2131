#!/usr/bin/env python3
"""How far ONE discrete gate flip moves the conv-stack gradients of tests/test_engine_gpu.py::test_full_geometry_logits_and_step.
The oracle (fp64) is re-run with every weight multiplied by (1 +- 6e-8), i.e. perturbed at fp32 rounding level.  Most trials move
every gradient by ~3e-6 (no ReLU gate / pool arg-max changes side); a trial in which one does moves conv1W by 8e-3, conv1b by
6e-3, conv2W/b by 4-5e-3, conv3b by 3e-3 and leaves the tensors above the flip untouched.  An fp32 kernel with a different (equally
valid) summation order is such a perturbation, so the test bounds the conv-stack tensors by a few flips, not by rounding error.
CPU only; ~2 minutes."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import lrcn_oracle as O
MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)
rng = np.random.default_rng(1)
shape, ncls, fpc, b = (227, 227, 3), 101, 4, 2
p = O.init_params(rng, ncls, "fc6", 256, 1, shape, classifier="lstm", well_scaled=True)
frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
lab = rng.integers(0, ncls, b)
onehot = O.labels_to_one_hot([[l] for l in lab], ncls)
x = frames.astype(np.float32) - MEAN
t0 = time.time()
_, loss, gn, acc, logits, g64 = O.lrcn_train_step(p, x, onehot, fpc, lr=1e-3, clip_norm=10.0, chunk=4)
print('fp64 %.1fs' % (time.time() - t0), flush=True)
def rel(g, k): return np.linalg.norm((g[k] - g64[k]).ravel()) / np.linalg.norm(g64[k].ravel())
keys = ["dcnn/conv1W", "dcnn/conv1b", "dcnn/conv2W", "dcnn/conv2b", "dcnn/conv3b", "dcnn/fc6W"]
_, _, _, _, _, g32 = O.lrcn_train_step(p, x, onehot, fpc, lr=1e-3, clip_norm=10.0, chunk=4, dtype=np.float32)
print('fp32 oracle   ', {k: '%.2e' % rel(g32, k) for k in keys}, flush=True)
prng = np.random.default_rng(7)
for trial in range(5):
    pp = {k: (v.astype(np.float64) * (1 + prng.uniform(-6e-8, 6e-8, v.shape))) for k, v in p.items()}
    _, _, _, _, _, g = O.lrcn_train_step(pp, x, onehot, fpc, lr=1e-3, clip_norm=10.0, chunk=4)
    print('fp64, weights * (1 +- 6e-8) trial %d' % trial, {k: '%.2e' % rel(g, k) for k in keys}, flush=True)
