#!/usr/bin/env python3
"""Generates tests/golden/graph_full.npz: the CPU oracle's answer (oracle.lrcn_oracle.model_forward / model_backward, fp64) for
multi-pipeline models at AlexNet's real layer shapes -- tests/graph_cases.py FULL_CASES:

  c4_ws   BASELINE config 4, the video-description encoder-decoder: 8 clips x 16 frames of 227x227 through AlexNet(fc6) + LSTM(256,
          fusion state) => the state of a 256-unit LSTM over 21 word vectors (300-d), per-step logits over a 1000-word vocabulary
  ts_ws   the two-stream LRCN: two dcnn towers over 4 clips x 16 frames each, fc6 features averaged into LSTM(256)

Parameters: vltf_amd.graph.init_params_for(model_specs(...), seed, well_scaled=True) -- the tests regenerate them and the inputs
(graph_cases.inputs) from the same seeds; only the oracle's outputs are stored (logits as float32, loss / global gradient norm /
accuracy, per-tensor gradient norms, 16-element heads and 64-element strided samples of every gradient, heads of the updated
parameters for lr 1e-3, clip_norm 10).  The reference ships no golden vectors and cannot run here (TensorFlow absent).

Run from the repo root (numpy fp64, a few minutes per case):  python tests/golden/make_golden_graph.py [case...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lrcn_oracle as O  # noqa: E402
from tests import graph_cases as GC  # noqa: E402

LR, CLIP = 1e-3, 10.0


def case_params(case):
    from vltf_amd.graph import init_params_for, model_specs
    pipes, ds = GC.specs_and_datasets(case)
    return init_params_for(model_specs(pipes, ds, case["V"]), seed=case["seed"], well_scaled=True)


def run_case(name, out):
    case = GC.FULL_CASES[name]()
    p = case_params(case)
    _, feeds = GC.inputs(case)
    t0 = time.time()
    logits, onehot, loss, grads, _ = GC.expect(case, p, feeds)
    clipped, gn = O.clip_by_global_norm(grads, CLIP)
    print("%s: %.0f s, loss %.6f, grad norm %.6f" % (name, time.time() - t0, loss, gn), flush=True)
    out[name + "/logits"] = logits.astype(np.float32)
    out[name + "/loss_gn_acc"] = np.array([loss, gn, O.accuracy(logits, onehot)])
    for k in sorted(p):
        g = grads[k].astype(np.float64).ravel()
        out["%s/gradnorm/%s" % (name, k)] = np.array([np.linalg.norm(g)])
        out["%s/gradhead/%s" % (name, k)] = g[:16].copy()
        idx = np.linspace(0, g.size - 1, 64).astype(np.int64)
        out["%s/gradsample/%s" % (name, k)] = g[idx].copy()
        newp = (p[k].astype(np.float64) - LR * clipped[k]).astype(np.float32)
        out["%s/newhead/%s" % (name, k)] = newp.ravel()[:16].astype(np.float64)


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_full.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in (sys.argv[1:] or list(GC.FULL_CASES)):
        for k in [k for k in out if k.startswith(name + "/")]:
            del out[k]
        run_case(name, out)
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
