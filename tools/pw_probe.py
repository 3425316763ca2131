#!/usr/bin/env python3
"""Runs one HBM-bound kernel of the path in isolation at a layer's real size (for rocprofv3 / timing).
usage: pw_probe.py <pool_lrn_bwd|lrn_pool_fwd|lrn_fwd|maxpool_fwd|maxpool_bwd|lrn_bwd|bias_grad> <1|2> [frames] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vltf_amd import ops

SHAPES = {"1": (96, 57, 57, 2), "2": (256, 28, 28, 1), "5": (256, 13, 13, 0)}   # c, h, w, pool-output halo


def main():
    what, layer = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    c, h, w, ph = SHAPES[layer]
    oh, ow = ops.pool_out(h), ops.pool_out(w)
    dev = "cuda:0"
    x = torch.rand(n, c, h, w, device=dev)
    y = torch.empty_like(x)
    dy = torch.rand(n, c, h, w, device=dev)
    p = torch.zeros(n, c, oh + 2 * ph, ow + 2 * ph, device=dev)
    dp = torch.rand_like(p)
    arg = torch.randint(0, 9, p.shape, dtype=torch.uint8, device=dev)
    ws = torch.empty(64 * c, device=dev)
    db = torch.empty(c, device=dev)
    dh = {"1": 0, "2": 2, "5": 1}[layer]                              # halo of the gradient the engine's layer writes (dy_halo)
    dxh = torch.zeros(n, c, h + 2 * dh, w + 2 * dh, device=dev)
    fns = {
        "pool_lrn_bwd": lambda: ops.pool_lrn_bwd(x, dp, arg, dxh, p_halo=ph, dx_halo=dh, relu_fused=True),
        "lrn_pool_fwd": lambda: ops.lrn_pool_fwd(x, p, arg, p_halo=ph),
        "lrn_fwd": lambda: ops.lrn_fwd(x, y),
        "lrn_bwd": lambda: ops.lrn_bwd(x, dy, y, relu_fused=True),
        "maxpool_fwd": lambda: ops.maxpool_fwd(x, p, arg, y_halo=ph),
        "maxpool_bwd": lambda: ops.maxpool_bwd(dp, arg, y, dy_halo=ph),
        "bias_grad": lambda: ops.bias_grad_nchw(dy, db, ws),
    }
    fn = fns[what]
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    gb = x.numel() * 4 / 1e9
    print("%s layer %s n=%d: %.3f ms  (%.2f GB tensor -> %.2f TB/s per tensor pass)" % (what, layer, n, dt * 1e3, gb, gb / dt / 1e3))


if __name__ == "__main__":
    main()
