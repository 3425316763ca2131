#!/usr/bin/env python3
"""The two-stream LRCN (models/model.py:18-162: a dcnn feature pipeline on `main`, one on `aux`, a third pipeline averaging their fc6
features into LSTM(256) -> 101 classes) at the benchmark's clip shape on one GPU: one clipped-SGD train step of
vltf_amd.graph.GraphEngine.  Prints one JSON line (clips/s; not the headline metric of bench.py: two towers = twice the conv work per clip).
usage: bench_graph.py [clips] [steps] [fusion: avg | maximum | concat] [conv_math: f32 | bf16x3 | bf16]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vltf_amd.graph import DatasetInfo, GraphEngine, PipelineSpec

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    fusion = sys.argv[3] if len(sys.argv) > 3 else "avg"
    math = sys.argv[4] if len(sys.argv) > 4 else "f32"
    V, T, H = 101, 16, 256
    dev = "cuda:0"
    pipes = [PipelineSpec("rgb", ["main"], "dcnn", frame_encoding_layer="fc6"),
             PipelineSpec("flow", ["aux"], "dcnn", frame_encoding_layer="fc6"),
             PipelineSpec("fuse", ["rgb", "flow"], "nop", classifier="lstm", lstm_params=(H, 1, "avg"), input_fusion=fusion)]
    data = {t: DatasetInfo("video", T, 1, clips, image_shape=(227, 227, 3)) for t in ("main", "aux")}
    eng = GraphEngine(pipes, data, V, device=dev, dropout_keep_prob=0.5, conv_math=math)
    eng.load_params(eng.init_params(seed=2))
    rng = np.random.default_rng(0)
    feeds = {t: dict(frames_u8=torch.from_numpy(rng.integers(0, 256, (clips * T, 227, 227, 3), dtype=np.uint8)).to(dev), mean_bgr=MEAN)
             for t in ("main", "aux")}
    onehot = torch.zeros((clips, V), dtype=torch.int32)
    onehot[torch.arange(clips), torch.from_numpy(rng.integers(0, V, clips))] = 1
    onehot = onehot.to(dev)
    for _ in range(3):
        eng.train_step(feeds, onehot, lr=1e-3, clip_norm=10.0, fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(feeds, onehot, lr=1e-3, clip_norm=10.0, fetch=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = eng.train_step(feeds, onehot, lr=1e-3, clip_norm=10.0, fetch=True)
    print(json.dumps({"workload": "two-stream LRCN: 2 x AlexNet(fc6) towers, input_fusion %s -> LSTM(256) -> %d classes, %d clips x %d frames "
                                  "227x227 per stream, full train step (GraphEngine, 3 pipelines)" % (fusion, V, clips, T),
                      "clips_per_s": round(clips / dt, 2), "ms_per_step": round(dt * 1e3, 3), "parameters": int(eng.w.numel()),
                      "dtype": "f32" if math == "f32" else math + " conv path, fp32 elsewhere", "data": "synthetic",
                      "check": {"loss": round(out["loss"], 4), "grad_norm": round(out["grad_norm"], 3), "rows": out["rows"]}}))


if __name__ == "__main__":
    main()
