#!/usr/bin/env python3
"""Checks of the LDS-DMA wgrad kernel on the benchmark layers: bitwise run-to-run determinism (3 runs, workspace
poisoned with NaN in between) and agreement with the bounds-tested dense-layout kernel (a different kernel, tile
shape and split).  usage: wgrad_ab.py [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vltf_amd import ops

GEOM = {"conv1": (3, 227, 227, 96, 11, 4, 1), "conv2": (96, 28, 28, 256, 5, 1, 2), "conv3": (256, 13, 13, 384, 3, 1, 1),
        "conv4": (384, 13, 13, 384, 3, 1, 2), "conv5": (384, 13, 13, 256, 3, 1, 2)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
    dev = "cuda:0"
    torch.manual_seed(0)
    for layer, (cin, h, w, cout, k, s, g) in GEOM.items():
        conv = ops.Conv(cin, h, w, cout, k, k, s, g)
        xh = conv.same_pad()
        dyh = conv.same_pad() if s == 1 else 0
        conv.set_halo(xh, 0, dyh, 0)
        xd = torch.randn(n, cin, h, w, device=dev)
        dyd = torch.randn(n, cout, conv.oh, conv.ow, device=dev)
        x = torch.zeros(n, cin, h + 2 * xh, w + 2 * xh, device=dev)
        x[:, :, xh:xh + h, xh:xh + w] = xd
        dy = torch.zeros(n, cout, conv.oh + 2 * dyh, conv.ow + 2 * dyh, device=dev)
        dy[:, :, dyh:dyh + conv.oh, dyh:dyh + conv.ow] = dyd
        ws = torch.empty(max(conv.wgrad_ws_bytes(n) // 4, 1), device=dev)
        outs = []
        for _ in range(3):
            dw = torch.full((k, k, cin // g, cout), 3.0, device=dev)
            db = torch.full((cout,), 3.0, device=dev)
            ws.fill_(float("nan"))
            if conv.fuses_bias():
                conv.wgrad(x, dy, dw, ws, db=db)
            else:
                conv.wgrad(x, dy, dw, ws)
                db = dyd.sum(dim=(0, 2, 3))
            torch.cuda.synchronize()
            outs.append((dw.clone(), db.clone()))
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        dense = ops.Conv(cin, h, w, cout, k, k, s, g)                   # halo 0: the bounds-tested kernel
        ws2 = torch.empty(max(dense.wgrad_ws_bytes(n) // 4, 1), device=dev)
        dw2 = torch.empty_like(outs[0][0])
        dense.wgrad(xd, dyd, dw2, ws2)
        rel = ((outs[0][0] - dw2).norm() / dw2.norm()).item()
        relb = ((outs[0][1] - dyd.sum(dim=(0, 2, 3))).norm() / dyd.sum(dim=(0, 2, 3)).norm()).item()
        print("%s n=%d: run-to-run bitwise %s; vs dense-layout kernel rel L2 dW %.2e; db vs torch sum %.2e; finite %s"
              % (layer, n, same, rel, relb, bool(torch.isfinite(outs[0][0]).all())))
        assert same and rel < 1e-5 and relb < 1e-5


if __name__ == "__main__":
    main()
