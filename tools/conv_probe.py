#!/usr/bin/env python3
"""Runs one AlexNet conv layer's fwd / dgrad / wgrad kernels in isolation (for rocprofv3 / timing).
usage: conv_probe.py <conv1..conv5> <fwd|dgrad|wgrad> [frames] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vltf_amd import ops

GEOM = {  # cin, h, w, cout, k, stride, groups
    "conv1": (3, 227, 227, 96, 11, 4, 1), "conv2": (96, 28, 28, 256, 5, 1, 2), "conv3": (256, 13, 13, 384, 3, 1, 1),
    "conv4": (384, 13, 13, 384, 3, 1, 2), "conv5": (384, 13, 13, 256, 3, 1, 2),
    # dense layers as 1x1 convolutions over ONE image whose pixels are the frames (activations stored [features][frames]):
    # run with frames = 1; the image width is the frame count
    "fc6x1024": (9216, 1, 1024, 4096, 1, 1, 1), "fc6x128": (9216, 1, 128, 4096, 1, 1, 1), "gxx1024": (4096, 1, 1024, 1024, 1, 1, 1)}
MACS = {"conv1": 113221152, "conv2": 240844800, "conv3": 149520384, "conv4": 112140288, "conv5": 74760192,
        "fc6x1024": 9216 * 4096 * 1024, "fc6x128": 9216 * 4096 * 128, "gxx1024": 4096 * 1024 * 1024}


def main():
    layer, what = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    cin, h, w, cout, k, s, g = GEOM[layer]
    conv = ops.Conv(cin, h, w, cout, k, k, s, g)
    dev = "cuda:0"
    torch.manual_seed(0)
    padded = os.environ.get("VL_PROBE_DENSE") is None          # default: the engine's zero-halo layout
    xh = conv.same_pad() if padded else 0
    dyh = conv.same_pad() if (padded and s == 1) else 0
    yh = int(os.environ.get("VL_PROBE_YHALO", "0"))             # halo of the OUTPUT (y of fwd, dx of dgrad): the engine's 13 x 13 layers write
    dxh = int(os.environ.get("VL_PROBE_DXHALO", "0"))           # into halo layouts (y3, y4: 1; dp2, dy3, dy4: 1; dp1: 2)
    conv.set_halo(xh, yh, dyh, dxh)
    if s > 1 and padded and os.environ.get("VL_PROBE_NO_PHASE") is None:
        conv.set_x_phase_split(True)            # the engine's layout of a strided conv's input (vl_conv_set_x_phase_split)

    def haloed(c_, h_, w_, halo):
        t = torch.zeros(n, c_, h_ + 2 * halo, w_ + 2 * halo, device=dev)
        t[:, :, halo:halo + h_, halo:halo + w_] = torch.randn(n, c_, h_, w_, device=dev)
        return t

    x = haloed(cin, h, w, xh)
    if getattr(conv, "x_phase", 1) > 1:         # same values, column-phase-split storage: [n, c * phase, H + 2 halo, ceil((W + 2 halo) / phase)]
        ph = conv.x_phase
        xs = torch.zeros(conv.x_shape(n), device=dev)
        for q in range(ph):
            cols = x[:, :, :, q::ph]
            xs[:, q::ph, :, :cols.shape[3]] = cols
        x = xs
    wt = torch.randn(k, k, cin // g, cout, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    y = torch.zeros(n, cout, conv.oh + 2 * yh, conv.ow + 2 * yh, device=dev)
    dy = haloed(cout, conv.oh, conv.ow, dyh)
    dx = torch.zeros(n, cin, h + 2 * dxh, w + 2 * dxh, device=dev)
    dw = torch.empty_like(wt)
    wtt = torch.empty(wt.numel(), device=dev)
    ws = torch.empty(max(conv.wgrad_ws_bytes(n) // 4, 1), device=dev)
    if what == "dgrad":
        conv.wt_transpose(wt, wtt)

    def run():
        if what == "fwd":
            conv.fwd(x, wt, b, y, relu=True)
        elif what == "dgrad":
            conv.dgrad(dy, wtt, dx)
        else:
            conv.wgrad(x, dy, dw, ws)

    run()
    torch.cuda.synchronize()
    if os.environ.get("VL_PROBE_SAVE"):            # result of this arithmetic, to compare runs under different VL_CONV_MATH
        torch.save({"fwd": y, "dgrad": dx, "wgrad": dw}[what].cpu(), os.environ["VL_PROBE_SAVE"])
    if os.environ.get("VL_PROBE_CMP"):
        ref = torch.load(os.environ["VL_PROBE_CMP"], weights_only=True)
        got = {"fwd": y, "dgrad": dx, "wgrad": dw}[what].cpu()
        print("  vs %s: rel L2 %.3e  max abs %.3e (ref max %.3e)" % (os.environ["VL_PROBE_CMP"], float((got - ref).norm() / ref.norm()),
                                                                  float((got - ref).abs().max()), float(ref.abs().max())))
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%s.%s n=%d: %.3f ms  %.1f TFLOP/s" % (layer, what, n, dt * 1e3, 2 * MACS[layer] * n / dt / 1e12))


if __name__ == "__main__":
    main()
