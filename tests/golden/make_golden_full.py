#!/usr/bin/env python3
"""Generates tests/golden/lrcn_full.npz: the CPU oracle's answer for the BENCHMARK job itself and for a 32-frame job.

Cases (full AlexNet geometry, 227x227x3, fc6 encode -> LSTM(256, 1 layer, avg) -> 101 classes, lr 1e-3, clip_norm 10):
  cfg2_ws    BASELINE config 2 = bench.py's workload: 64 clips x 16 frames, well-scaled weights (sigma = sqrt(2/fan_in))
  cfg2_ref   the same job with the reference initialiser (sigma 0.05, alexnet.py:40-46) -- exactly bench.py's inputs:
             frames default_rng(0), labels default_rng(1000), parameters engine.init_params(cfg, seed=2)
  t32_ws     BASELINE config 5's clip length: 4 clips x 32 frames, well-scaled weights
  c3shard_ref  BASELINE config 3: rank 0's shard of bench.py's strong-scaling job at 8 GPUs -- clips 0..7 of the 64 (8 clips x 16
             frames, the first 128 frames / 8 labels of cfg2_ref's streams), reference initialiser; a plain 8-clip step (the
             shard's 1/64 loss scaling is a constant factor 8/64 on every gradient, applied by the test)
  c3pair_ws  the same 8 clips as ONE global batch with well-scaled weights: the answer a 2-rank run (4 + 4 clips) must reproduce
  t32_ws_q   t32_ws evaluated with the operand roundings of the engine's bf16 conv path (oracle q = bf16_round: every tensor that
             path stores as packed bf16 is rounded to bf16, accumulation stays fp64) -- the arithmetic BASELINE config 5 runs in

Inputs are regenerated from these seeds by the tests (tests/test_full_workload_gpu.py) and by bench.py's first-step check;
only the oracle's outputs are stored: logits, loss, global gradient norm, accuracy, per-tensor gradient norms, 16-element heads
of every gradient and updated parameter, and per-layer activation norms (to localise a disagreement).

The reference ships no golden vectors and cannot run here (TensorFlow absent): this pins the HIP path to the
oracle at the size the benchmark runs, the oracle itself is pinned by tests/test_oracle.py (torch-CPU autograd).

Run from the repo root (numpy fp64; ~20 GB of memory, a few minutes per case):  python tests/golden/make_golden_full.py [case...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lrcn_oracle as O  # noqa: E402

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)
SHAPE, NCLS, HID = (227, 227, 3), 101, 256
LR, CLIP = 1e-3, 10.0
CASES = {
    # name: (clips, fpc, well_scaled)
    "cfg2_ws": (64, 16, True),
    "cfg2_ref": (64, 16, False),
    "t32_ws": (4, 32, True),
    "c3shard_ref": (8, 16, False),
    "c3pair_ws": (8, 16, True),
    "t32_ws_q": (4, 32, True, "bf16"),
}
ACT_LAYERS = ("conv1", "lrn1", "pool1", "conv2", "lrn2", "pool2", "conv3", "conv4", "conv5", "pool5", "fc6")


def case_inputs(name):
    """(params, frames u8 [clips*fpc,227,227,3], onehot int32 [clips,101]) -- bench.py's seeds (rank 0)."""
    from vltf_amd.engine import NetConfig, init_params      # numpy-only initialiser shared with bench.py
    clips, fpc, ws = CASES[name][:3]
    cfg = NetConfig(image_shape=SHAPE, num_classes=NCLS, fpc=fpc, lstm_hidden=HID)
    p = init_params(cfg, seed=2, well_scaled=ws)
    frames = np.random.default_rng(0).integers(0, 256, (clips * fpc,) + SHAPE, dtype=np.uint8)
    lab = np.random.default_rng(1000).integers(0, NCLS, clips)
    return p, frames, O.labels_to_one_hot([[l] for l in lab], NCLS)


def run_case(name, out):
    clips, fpc, ws = CASES[name][:3]
    q = O.bf16_round if len(CASES[name]) > 3 else None
    p, frames, onehot = case_inputs(name)
    x = frames.astype(np.float32) - MEAN
    t0 = time.time()
    logits, cache = O.lrcn_forward(p, x, fpc, keep=True, chunk=32, q=q)
    loss, dlogits = O.softmax_xent_mean(logits, onehot)
    print("%s: forward %.0f s, loss %.6f" % (name, time.time() - t0, loss), flush=True)
    for l in ACT_LAYERS:
        out["%s/actnorm/%s" % (name, l)] = np.array([np.sqrt(sum(float((c[l].astype(np.float64) ** 2).sum()) for c in cache["cnn"]))])
    hseq = cache["lstm"][0]
    out[name + "/actnorm/lstm_c_last"] = np.array([np.linalg.norm(hseq["cs"][-1])])
    out[name + "/fused"] = cache["fused"].astype(np.float64)
    grads = O.lrcn_backward(p, cache, dlogits, fpc, q=q)
    del cache
    clipped, gn = O.clip_by_global_norm(grads, CLIP)
    print("%s: backward done %.0f s, grad norm %.6f" % (name, time.time() - t0, gn), flush=True)
    out[name + "/logits"] = logits.astype(np.float64)
    out[name + "/loss_gn_acc"] = np.array([loss, gn, O.accuracy(logits, onehot)])
    for k in sorted(p):
        g = grads[k].astype(np.float64).ravel()
        out["%s/gradnorm/%s" % (name, k)] = np.array([np.linalg.norm(g)])
        out["%s/gradhead/%s" % (name, k)] = g[:16].copy()
        # a strided sample across the whole tensor: 64 elements
        idx = np.linspace(0, g.size - 1, 64).astype(np.int64)
        out["%s/gradsample/%s" % (name, k)] = g[idx].copy()
        newp = (p[k].astype(np.float64) - LR * clipped[k]).astype(np.float32)
        out["%s/newhead/%s" % (name, k)] = newp.ravel()[:16].astype(np.float64)


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lrcn_full.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in (sys.argv[1:] or list(CASES)):
        for k in [k for k in out if k.startswith(name + "/")]:
            del out[k]
        run_case(name, out)
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
