#!/usr/bin/env python3
"""Times pool5's forward / backward ((h, w, c)-flat pooled tensors, 13 x 13 x 256 planes, halo-1 dx with ReluGrad mask) in isolation.
usage: pool5_probe.py [frames] [iters]     (VL_MAXPOOL_GENERIC=1: the element-per-thread backward kernel, A/B)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vltf_amd import ops


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = "cuda:0"
    c, h, w = 256, 13, 13
    x = torch.relu(torch.randn(n, c, h, w, device=dev))
    p = torch.empty(n, 6, 6, c, device=dev)
    arg = torch.empty(n, 6, 6, c, dtype=torch.uint8, device=dev)
    dp = torch.randn(n, 6, 6, c, device=dev)
    dx = torch.zeros(n, c, h + 2, w + 2, device=dev)
    cases = {"maxpool_fwd hwc": (lambda: ops.maxpool_fwd(x, p, arg, hwc=True), 4.0 * n * c * (h * w + 36) + n * c * 36),
             "maxpool_bwd hwc + mask": (lambda: ops.maxpool_bwd(dp, arg, dx, relu_mask=x, hwc=True, dx_halo=1), 4.0 * n * c * (2 * h * w + 36) + n * c * 36)}
    for name, (fn, nbytes) in cases.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print("%-24s frames %d: %.3f ms  %.2f TB/s algorithmic" % (name, n, ms, nbytes / ms * 1e-9))


if __name__ == "__main__":
    main()
