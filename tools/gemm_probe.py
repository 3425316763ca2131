#!/usr/bin/env python3
"""Times the dense GEMMs of one LRCN train step (fc6 forward / weight gradient / input gradient, the LSTM projections) in
isolation at a given frame count.  usage: gemm_probe.py [frames] [iters]      (VL_GEMM_WANT=<workgroups per CU> to experiment)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vltf_amd import ops


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = "cuda:0"
    F, D, H4 = 9216, 4096, 1024
    ws = torch.empty(64 << 20, device=dev)
    p5, w6, f6 = torch.randn(n, F, device=dev), torch.randn(F, D, device=dev) * 0.01, torch.empty(n, D, device=dev)
    b6 = torch.zeros(D, device=dev)
    d6, g6, dp5 = torch.randn(n, D, device=dev), torch.empty(F, D, device=dev), torch.empty(n, F, device=dev)
    kx, gx, dz = torch.randn(D + 256, H4, device=dev) * 0.01, torch.empty(n, H4, device=dev), torch.randn(n, H4, device=dev)
    gk, hp = torch.empty(D + 256, H4, device=dev), torch.randn(n, 256, device=dev)
    cases = {
        "fc6.fwd   [n,9216]x[9216,4096]": (lambda: ops.gemm(p5, w6, f6, n, D, F, bias=b6, relu=True, ws=ws), 2.0 * n * D * F, 4.0 * F * D),
        "fc6.wgrad [9216,n]x[n,4096]": (lambda: ops.gemm(p5, d6, g6, F, D, n, transa=True, ws=ws), 2.0 * n * D * F, 4.0 * F * D),
        "fc6.dgrad [n,4096]x[4096,9216]T": (lambda: ops.gemm(d6, w6, dp5, n, F, D, transb=True, ws=ws), 2.0 * n * D * F, 4.0 * F * D),
        "lstm.gx   [n,4096]x[4096,1024]": (lambda: ops.gemm(f6, kx, gx, n, H4, D, bias=None, ws=ws), 2.0 * n * D * H4, 4.0 * D * H4),
        "lstm.dK   [4096,n]x[n,1024]": (lambda: ops.gemm(f6, dz, gk, D, H4, n, transa=True, ws=ws), 2.0 * n * D * H4, 4.0 * D * H4),
        "lstm.dKh  [256,n]x[n,1024]": (lambda: ops.gemm(hp, dz, gk[D:], 256, H4, n, transa=True, ws=ws), 2.0 * n * 256 * H4, 4.0 * 256 * H4),
        "lstm.dx   [n,1024]x[1024,4096]T": (lambda: ops.gemm(dz, kx, d6, n, D, H4, transb=True, ldb=H4, relu_mask=f6, ws=ws), 2.0 * n * D * H4, 4.0 * D * H4),
    }
    tot = 0.0
    for name, (fn, flop, wbytes) in cases.items():
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        tot += dt
        print("%-34s n=%d: %7.1f us  %6.1f TFLOP/s  weights at %5.2f TB/s" % (name, n, dt * 1e6, flop / dt / 1e12, wbytes / dt / 1e12))
    print("total %.1f us (VL_GEMM_WANT=%s)" % (tot * 1e6, os.environ.get("VL_GEMM_WANT", "3")))


if __name__ == "__main__":
    main()
