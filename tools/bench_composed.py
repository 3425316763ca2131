#!/usr/bin/env python3
"""BASELINE config 4 at its full shape on one GPU: AlexNet(fc6) + LSTM(256, state) encoder over 16-frame 227x227 clips feeding a 256-unit
decoder LSTM over 21 word vectors (BOS + 20 tokens, 300-d) with per-step logits over a 1000-word vocabulary; one clipped-SGD train step
(vltf_amd.composed.ComposedEngine).  Prints one JSON line (clips/s; not the headline metric of bench.py).
usage: bench_composed.py [clips] [steps] [conv_math: f32 | bf16x3 | bf16]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vltf_amd.composed import ComposedEngine, HeadConfig, init_head_params
from vltf_amd.engine import NetConfig, init_params

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    math = sys.argv[3] if len(sys.argv) > 3 else "f32"
    V, E, Tf, Tw, H = 1000, 300, 16, 21, 256
    dev = "cuda:0"
    enc = NetConfig(image_shape=(227, 227, 3), num_classes=V, fpc=Tf, classifier="lstm", lstm_hidden=H, lstm_layers=1, fusion="state",
                    dropout_keep_prob=0.5, conv_math=math)
    head = HeadConfig(in_dim=E, fpc=Tw, num_classes=V, lstm_hidden=H, lstm_layers=1, fusion="reshape", dropout_keep_prob=0.5)
    eng = ComposedEngine(enc, head, max_clips=clips, device=dev)
    p = {"enc/" + k: v for k, v in init_params(enc, seed=2).items()}
    p.update(init_head_params(enc, head, seed=3))
    eng.load_params(p)
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, (clips * Tf, 227, 227, 3), dtype=np.uint8)).to(dev)
    words = torch.from_numpy(rng.standard_normal((clips * Tw, E)).astype(np.float32)).to(dev)
    onehot = torch.zeros((clips * Tw, V), dtype=torch.int32)
    onehot[torch.arange(clips * Tw), torch.from_numpy(rng.integers(0, V, clips * Tw))] = 1
    onehot = onehot.to(dev)
    for _ in range(3):
        eng.train_step(frames, words, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN, fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(frames, words, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN, fetch=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = eng.train_step(frames, words, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN, fetch=True)
    print(json.dumps({"workload": "config 4: AlexNet(fc6)+LSTM(256,state) encoder -> LSTM(256) decoder, %d clips x %d frames 227x227 + %d word "
                                  "vectors (%d-d), %d classes, full train step" % (clips, Tf, Tw, E, V),
                      "clips_per_s": round(clips / dt, 2), "ms_per_step": round(dt * 1e3, 3), "dtype": "f32" if math == "f32" else math + " conv path, fp32 elsewhere", "data": "synthetic",
                      "check": {"loss": round(out["loss"], 4), "grad_norm": round(out["grad_norm"], 3), "rows": out["rows"]}}))


if __name__ == "__main__":
    main()
