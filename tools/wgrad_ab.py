#!/usr/bin/env python3
"""A/B of the two padded-mode wgrad kernels on the benchmark layers: bitwise run-to-run determinism of the LDS-DMA form
and its agreement with the register-staged form (VL_WGRAD_1BUF=1).  usage: wgrad_ab.py [frames]"""
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
        x = torch.zeros(n, cin, h + 2 * xh, w + 2 * xh, device=dev)
        x[:, :, xh:xh + h, xh:xh + w] = torch.randn(n, cin, h, w, device=dev)
        dy = torch.zeros(n, cout, conv.oh + 2 * dyh, conv.ow + 2 * dyh, device=dev)
        dy[:, :, dyh:dyh + conv.oh, dyh:dyh + conv.ow] = torch.randn(n, cout, conv.oh, conv.ow, device=dev)
        ws = torch.empty(max(conv.wgrad_ws_bytes(n) // 4, 1), device=dev)
        outs = []
        for mode in ("dma", "dma", "dma", "1buf"):
            if mode == "1buf":
                os.environ["VL_WGRAD_1BUF"] = "1"
            else:
                os.environ.pop("VL_WGRAD_1BUF", None)
            dw = torch.full((k, k, cin // g, cout), 3.0, device=dev)
            db = torch.full((cout,), 3.0, device=dev)
            ws.fill_(float("nan"))
            if conv.fuses_bias():
                conv.wgrad(x, dy, dw, ws, db=db)
            else:
                conv.wgrad(x, dy, dw, ws)
                db = dy.sum(dim=(0, 2, 3))
            torch.cuda.synchronize()
            outs.append((dw.clone(), db.clone()))
        os.environ.pop("VL_WGRAD_1BUF", None)
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:3])
        rel = ((outs[0][0] - outs[3][0]).norm() / outs[3][0].norm()).item()
        relb = ((outs[0][1] - outs[3][1]).norm() / outs[3][1].norm()).item()
        ref = torch.nn.grad.conv2d_weight(x[:, :, xh - conv.pad_t:, xh - conv.pad_l:] if False else x, (cout, cin // g, k, k), dy) if False else None
        print("%s n=%d: dma run-to-run bitwise %s; dma vs 1buf rel L2 dW %.2e db %.2e; finite %s"
              % (layer, n, same, rel, relb, bool(torch.isfinite(outs[0][0]).all())))
        assert same and rel < 1e-5 and relb < 1e-5


if __name__ == "__main__":
    main()
