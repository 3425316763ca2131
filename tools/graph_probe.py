#!/usr/bin/env python3
"""Is the small-batch train step launch-bound?  Times train_step_u8 eagerly and as a captured hipGraph replay.
usage: graph_probe.py [clips] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vltf_amd.engine import LRCNEngine, NetConfig, init_params

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = "cuda:0"
    cfg = NetConfig(image_shape=(227, 227, 3), num_classes=101, fpc=16, classifier="lstm", lstm_hidden=256, lstm_layers=1, dropout_keep_prob=0.5)
    eng = LRCNEngine(cfg, max_clips=clips, device=dev)
    eng.load_params(init_params(cfg, seed=2))
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, (clips * 16, 227, 227, 3), dtype=np.uint8)).to(dev)
    onehot = torch.zeros((clips, 101), dtype=torch.int32)
    onehot[torch.arange(clips), torch.from_numpy(rng.integers(0, 101, clips))] = 1
    onehot = onehot.to(dev)
    step = lambda: eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN, fetch=False)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    # host-side cost of issuing one step (no GPU wait): how far ahead of the GPU the host can run
    t0 = time.perf_counter()
    step()
    issue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    graph_ms = None
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / steps * 1e3
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
    print("clips %d: eager %.3f ms/step, host issue time of one step %.3f ms, graph replay %s ms/step" % (clips, eager, issue,
          "%.3f" % graph_ms if graph_ms else "n/a"))


if __name__ == "__main__":
    main()
