#!/usr/bin/env python3
"""conv1 on the bf16 path (3x3 stride-1 layer over the 48-channel space-to-depth input): where does the time go?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vltf_amd.ops as ops

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
conv = ops.Conv(3, 227, 227, 96, 11, 11, 4, 1)
conv.set_halo(conv.same_pad(), 0, 0, 0)
eq = conv.s2d_layer()
xb = torch.randn(ops.c8_shape(n, 48, 57, 57, 1), device=dev).bfloat16()
w = torch.randn(eq.w_shape, device=dev) * 0.05
wb = torch.zeros(eq.c8_w_bytes(False), dtype=torch.uint8, device=dev)
eq.c8_pack_w(w, wb, False)
b = torch.zeros(96, device=dev)
y = torch.zeros((n, 96, 57, 57), device=dev)
flops = 2.0 * n * 57 * 57 * 96 * 9 * 48
print("fp32 dense y only : %.3f ms" % timed(lambda: eq.c8_fwd(xb, wb, b, y=y)))
eq.set_halo(1, 1, 1, 0)
yb = torch.zeros(ops.c8_shape(n, 96, 57, 57, 1), dtype=torch.bfloat16, device=dev)
yh = torch.zeros((n, 96, 59, 59), device=dev)
print("packed yb only    : %.3f ms" % timed(lambda: eq.c8_fwd(xb, wb, b, yb=yb)))
print("fp32 haloed y only: %.3f ms (scalar stores)" % timed(lambda: eq.c8_fwd(xb, wb, b, y=yh)))
print("MFMA time at 2.5 PF: %.3f ms" % (flops / 2.5e15 * 1e3))
for cin in (48, 96, 192):
    c2 = ops.Conv(cin, 57, 57, 96, 3, 3, 1, 1)
    c2.set_halo(1, 1, 1, 0)
    x2 = torch.randn(ops.c8_shape(n, cin, 57, 57, 1), device=dev).bfloat16()
    w2 = torch.randn(c2.w_shape, device=dev) * 0.05
    wb2 = torch.zeros(c2.c8_w_bytes(False), dtype=torch.uint8, device=dev)
    c2.c8_pack_w(w2, wb2, False)
    print("cin %3d (%2d stages): packed out %.3f ms" % (cin, cin // 8 * 9 // 4 + (1 if cin // 8 * 9 % 4 else 0), timed(lambda: c2.c8_fwd(x2, wb2, b, yb=yb))))
