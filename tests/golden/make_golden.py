#!/usr/bin/env python3
"""Generates tests/golden/lrcn_small.npz from the CPU oracle (oracle/lrcn_oracle.py).

The reference ships no golden vectors and cannot run here (TensorFlow absent): these fixtures pin the
ORACLE (and through it the HIP path) against silent regressions; the oracle itself is pinned by the
torch-CPU cross-checks in tests/test_oracle.py.  Inputs are regenerated from seeds by the tests, only the
expected outputs are stored.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lrcn_oracle as O  # noqa: E402

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)
CASES = {
    # name: (image_shape, classes, fpc, clips, final_layer, hidden, layers, fusion, seed)
    "fc6_lstm8_avg": ((67, 67, 3), 7, 3, 2, "fc6", 8, 1, "avg", 11),
    "fc7_lstm12x2_last": ((67, 67, 3), 5, 2, 3, "fc7", 12, 2, "last", 12),
}


def case_inputs(name):
    shape, ncls, fpc, b, layer, hid, layers, fusion, seed = CASES[name]
    rng = np.random.default_rng(seed)
    p = O.init_params(rng, ncls, layer, hid, layers, shape, well_scaled=True)
    frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
    labels = rng.integers(0, ncls, b)
    return p, frames, O.labels_to_one_hot([[l] for l in labels], ncls)


def main():
    out = {}
    for name, (shape, ncls, fpc, b, layer, hid, layers, fusion, seed) in CASES.items():
        p, frames, onehot = case_inputs(name)
        x = frames.astype(np.float32) - MEAN
        newp, loss, gn, acc, logits, grads = O.lrcn_train_step(p, x, onehot, fpc, lr=0.01, clip_norm=0.5, final_layer=layer,
                                                              lstm_layers=layers, fusion=fusion)
        out[name + "/logits"] = logits.astype(np.float64)
        out[name + "/loss_gn_acc"] = np.array([loss, gn, acc])
        for k in sorted(p):
            g = grads[k].astype(np.float64).ravel()
            out[name + "/gradnorm/" + k] = np.array([np.linalg.norm(g)])
            out[name + "/gradhead/" + k] = g[:16].copy()
            out[name + "/newhead/" + k] = newp[k].ravel()[:16].astype(np.float64)
    # op-level known answers
    rng = np.random.default_rng(99)
    x = np.maximum(rng.standard_normal((1, 5, 5, 8)) * 30, 0).astype(np.float32)
    out["lrn/x"], out["lrn/y"] = x, O.lrn(x)[0]
    y, arg = O.max_pool_valid(x)
    out["pool/y"], out["pool/arg"] = y, arg
    out["same_pad"] = np.array([O.same_pad(227, 11, 4), O.same_pad(224, 11, 4), O.same_pad(28, 5, 1), O.same_pad(13, 3, 1)])
    out["lr_table"] = np.array(O.precompute_learning_rates(0.05, ["exp", "drops", 4, 0.5], 5, 2))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lrcn_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
