#!/usr/bin/env python3
"""Times the packed-bf16 (c8) conv kernels on AlexNet's stride-1 layers: c8_probe.py [frames] [reps]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vltf_amd.ops as ops

LAYERS = {"conv2": (96, 27, 27, 256, 5, 2), "conv3": (256, 13, 13, 384, 3, 1), "conv4": (384, 13, 13, 384, 3, 2), "conv5": (384, 13, 13, 256, 3, 2)}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else list(LAYERS)
    dev = "cuda:0"
    for name in only:
        cin, h, w, cout, k, g = LAYERS[name]
        conv = ops.Conv(cin, h, w, cout, k, k, 1, g)
        pad = conv.same_pad()
        conv.set_halo(pad, pad, pad, pad)
        flops = 2.0 * n * h * w * cout * k * k * cin / g
        xb = torch.randn(ops.c8_shape(n, cin, h, w, pad), device=dev).bfloat16()
        dyb = torch.randn(ops.c8_shape(n, cout, h, w, pad), device=dev).bfloat16()
        wd = torch.randn(conv.w_shape, device=dev) * 0.05
        bd = torch.zeros(cout, device=dev)
        wb = torch.zeros(conv.c8_w_bytes(False), dtype=torch.uint8, device=dev)
        wbt = torch.zeros(conv.c8_w_bytes(True), dtype=torch.uint8, device=dev)
        conv.c8_pack_w(wd, wb, False)
        conv.c8_pack_w(wd, wbt, True)
        y = torch.zeros((n, cout, h + 2 * pad, w + 2 * pad), device=dev)
        yb = torch.zeros(ops.c8_shape(n, cout, h, w, pad), dtype=torch.bfloat16, device=dev)
        dx = torch.zeros((n, cin, h + 2 * pad, w + 2 * pad), device=dev)
        dxb = torch.zeros(ops.c8_shape(n, cin, h, w, pad), dtype=torch.bfloat16, device=dev)
        dw = torch.empty_like(wd)
        ws = torch.empty(max(conv.c8_wgrad_ws_bytes(n) // 4, 1), device=dev)
        res = {}
        res["fwd f32+c8"] = timed(lambda: conv.c8_fwd(xb, wb, bd, y=y, yb=yb), reps)
        res["fwd c8"] = timed(lambda: conv.c8_fwd(xb, wb, bd, yb=yb), reps)
        res["dgrad f32+c8"] = timed(lambda: conv.c8_dgrad(dyb, wbt, dx=dx, dxb=dxb), reps)
        res["dgrad c8"] = timed(lambda: conv.c8_dgrad(dyb, wbt, dxb=dxb), reps)
        res["wgrad"] = timed(lambda: conv.c8_wgrad(xb, dyb, dw, ws), reps)
        print(name, "frames", n, " ".join("%s %.3f ms (%.0f TF)" % (k_, v, flops / v * 1e-9) for k_, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
