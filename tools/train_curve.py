#!/usr/bin/env python3
"""Loss curves of the SAME job under the conv arithmetics: one fixed batch (the network memorises it), the benchmark's settings
(lr 1e-3, clip 10, SGD), `steps` steps from identical parameters.  usage: train_curve.py [clips] [steps] [init: scaled|reference]
Prints the loss at a few steps for f32, bf16x6, bf16x3, bf16.  GPU; for DESIGN.md 4.6."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import lrcn_oracle as O
from vltf_amd.engine import LRCNEngine, NetConfig, init_params

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    init = sys.argv[3] if len(sys.argv) > 3 else "scaled"
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, (clips * 16, 227, 227, 3), dtype=np.uint8)).to("cuda:0")
    onehot = torch.zeros((clips, 101), dtype=torch.int32)
    onehot[torch.arange(clips), torch.from_numpy(rng.integers(0, 101, clips))] = 1
    onehot = onehot.to("cuda:0")
    marks = sorted(set([1, 2, 5, 10, 20, 30, 40, 50, steps]) & set(range(1, steps + 1)))
    print("init %s, %d clips, lr 1e-3, clip 10; loss BEFORE the step named" % (init, clips))
    print("%-8s" % "math" + "".join("%10d" % m for m in marks) + "   accuracy at the end")
    for math in ("f32", "bf16x6", "bf16x3", "bf16"):
        cfg = NetConfig(image_shape=(227, 227, 3), num_classes=101, fpc=16, conv_math=math)
        eng = LRCNEngine(cfg, max_clips=clips, device="cuda:0")
        if init == "reference":
            eng.load_params(init_params(cfg, seed=2))
        else:
            eng.load_params(O.init_params(np.random.default_rng(2), 101, "fc6", 256, 1, (227, 227, 3), well_scaled=True))
        losses, acc = [], 0.0
        for _ in range(steps):
            o = eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN)
            losses.append(o["loss"])
            acc = o["accuracy"]
        print("%-8s" % math + "".join("%10.4f" % losses[m - 1] for m in marks) + "   %.3f" % acc, flush=True)
        del eng


if __name__ == "__main__":
    main()
