#!/usr/bin/env python3
"""Times the pool / LRN kernels that write packed bf16 (bf16 path) next to their fp32 forms at AlexNet's two shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vltf_amd.ops as ops

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
for (c, h, halo_p, halo_dx) in ((96, 57, 2, 1), (256, 28, 1, 2)):
    x = torch.relu(torch.randn((n, c, h, h), device=dev)) * 30
    oh = ops.pool_out(h)
    p = torch.zeros((n, c, oh + 2 * halo_p, oh + 2 * halo_p), device=dev)
    arg = torch.zeros(p.shape, dtype=torch.uint8, device=dev)
    pb = torch.zeros(ops.c8_shape(n, c, oh, oh, halo_p), dtype=torch.bfloat16, device=dev)
    dp = torch.randn(p.shape, device=dev)
    dx = torch.zeros((n, c, h + 2 * halo_dx, h + 2 * halo_dx), device=dev)
    dxb = torch.zeros(ops.c8_shape(n, c, h, h, halo_dx), dtype=torch.bfloat16, device=dev)
    gb = (x.numel() * 4 + p.numel() * 5) / 1e9
    t1 = timed(lambda: ops.lrn_pool_fwd(x, p, arg, p_halo=halo_p))
    t2 = timed(lambda: ops.lrn_pool_fwd_c8(x, pb, arg, p_halo=halo_p))
    t3 = timed(lambda: ops.pool_lrn_bwd(x, dp, arg, dx, p_halo=halo_p, dx_halo=halo_dx))
    t4 = timed(lambda: ops.pool_lrn_bwd_c8(x, dp, arg, dxb, p_halo=halo_p, dxb_halo=halo_dx))
    # the forms the bf16 path's step launches: x packed too (the conv's packed output, no halo)
    xb = torch.zeros(ops.c8_shape(n, c, h, h, 0), dtype=torch.bfloat16, device=dev)
    ops.pack_c8(x, xb, 0, 0)
    t5 = timed(lambda: ops.lrn_pool_fwd_c8(xb, pb, arg, p_halo=halo_p, channels=c))
    t6 = timed(lambda: ops.pool_lrn_bwd_c8(xb, dp, arg, dxb, p_halo=halo_p, dxb_halo=halo_dx))
    print("c %3d h %2d: lrn_pool_fwd %.3f ms  _c8 %.3f ms  packed x %.3f ms | pool_lrn_bwd %.3f ms  _c8 %.3f ms  packed x %.3f ms"
          % (c, h, t1, t2, t5, t3, t4, t6), flush=True)
