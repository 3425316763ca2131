"""Micro-reproducer of round 3's open defect: `pool_lrn_bwd_stream_kernel<5,3,true,0>` (conv2's pool / LRN backward) gives different
outputs on identical inputs while conv3's split-bf16 weight gradient runs on another stream.  Runs the pair in isolation, many times,
and classifies every differing element: lane within its wave, channel phase of the channel walk, and which quantity the wrong value
equals (host fp64 evaluation of the exact result, of the routed gradient g alone, ...).

usage: plb_race_probe.py [frames] [trials] [perturbers, comma separated: none,f32,bf16x3,bf16x6,fwd3,copy]
The library under test comes from VLTF_HIP_LIB (variant builds of tools/plb_variants.sh) or the in-tree one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vltf_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
PERT = (sys.argv[3] if len(sys.argv) > 3 else "none,f32,bf16x3").split(",")
DEV = "cuda:0"
torch.manual_seed(0)
C2, H2, OH = 256, 28, 13
ALPHA, BETA, BIAS = 2e-5, 0.75, 1.0

# ---- layer-2 tensors: y2 (conv2 output after ReLU), pooled output / arg-max (halo 1), pooled gradient, dy2 (halo 2)
y2 = torch.relu(torch.randn(N, C2, H2, H2, device=DEV) * 70.0).contiguous()
p2 = torch.zeros(N, C2, OH + 2, OH + 2, device=DEV)
arg2 = torch.zeros(N, C2, OH + 2, OH + 2, device=DEV, dtype=torch.uint8)
ops.lrn_pool_fwd(y2, p2, arg2, p_halo=1)
dp2 = torch.zeros_like(p2)
dp2[:, :, 1:-1, 1:-1] = torch.randn(N, C2, OH, OH, device=DEV) * 1e-6
dy2 = torch.zeros(N, C2, H2 + 4, H2 + 4, device=DEV)

# ---- conv3 (3x3, 256 -> 384 on 13 x 13) weight gradient: x = pooled output (halo 1), dy3 (halo 1)
conv3 = ops.Conv(256, 13, 13, 384, 3, 3, 1, 1)
conv3.set_halo(1, 1, 1, 1)
dy3 = torch.zeros(N, 384, 15, 15, device=DEV)
dy3[:, :, 1:-1, 1:-1] = torch.randn(N, 384, 13, 13, device=DEV) * 1e-3
dw3 = torch.zeros(3, 3, 256, 384, device=DEV)
db3 = torch.zeros(384, device=DEV)
w3 = (torch.randn(3, 3, 256, 384, device=DEV) * 0.05).contiguous()
b3 = torch.full((384,), 0.1, device=DEV)
y3 = torch.zeros(N, 384, 15, 15, device=DEV)
ws_bytes = 0
for m in ("f32", "bf16x3", "bf16x6"):
    ops.set_conv_math(m)
    ws_bytes = max(ws_bytes, conv3.wgrad_ws_bytes(N))
ws = torch.zeros(ws_bytes // 4 + 64, device=DEV)
big_a = torch.zeros(64 << 20, device=DEV)
big_b = torch.zeros(64 << 20, device=DEV)
side = torch.cuda.Stream()


def plb():
    ops.pool_lrn_bwd(y2, dp2, arg2, dy2, p_halo=1, dx_halo=2, relu_fused=True)


_spin = None


def spinner(bf16, grid, lds_bytes, iters):
    """tools/ubench/pk_f32_war.hip's MFMA spinner (registers only: no LDS traffic, no memory traffic) as the neighbour"""
    global _spin
    import ctypes as C
    if _spin is None:
        _spin = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libpkwar.so"))
        _spin.pkwar_spin.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
    _spin.pkwar_spin(bf16, grid, lds_bytes, iters, big_a.data_ptr(), torch.cuda.current_stream().cuda_stream)


def perturb(kind):
    if kind == "none":
        return
    if kind.startswith("spin"):           # spin-bf16-same | spin-f32-same | spin-bf16-other (90 KB of LDS each on half the CUs: never beside a
        _, ty, where = kind.split("-")    # pool_lrn_bwd workgroup only if that one is large too -- use with VL_PLB_EXTRA_LDS)
        spinner(int(ty == "bf16"), 256 if where == "same" else 128, 8192 if where == "same" else 150 * 1024, 3000)
        return
    if kind == "copy":
        big_b.copy_(big_a)
        return
    if kind == "fwd3":
        ops.set_conv_math("bf16x3")
        conv3.fwd(p2, w3, b3, y3)
        return
    ops.set_conv_math(kind)
    conv3.wgrad(p2, dy3, dw3, ws, db3 if conv3.fuses_bias() else None)


torch.cuda.synchronize()
plb()
torch.cuda.synchronize()
ref = dy2.clone()
alone_bad = 0
for _ in range(5):
    dy2.zero_()
    plb()
    torch.cuda.synchronize()
    alone_bad += int((dy2 != ref).sum())
print("library:", os.environ.get("VLTF_HIP_LIB", "in-tree"), " frames", N, " alone: differing elements over 5 repeats:", alone_bad, flush=True)

y2h = dp2h = arg2h = None


def host_terms(n, c, py, px):
    """fp64 evaluation of output (n, c, py, px): exact r, routed gradient g[c], u = g pw, and the LRN correction term."""
    global y2h, dp2h, arg2h
    if y2h is None:
        y2h, dp2h, arg2h = y2.cpu().numpy().astype(np.float64), dp2.cpu().numpy().astype(np.float64), arg2.cpu().numpy()

    def g_of(ch):
        if ch < 0 or ch >= C2:
            return 0.0
        g = 0.0
        for oh in (py >> 1, (py >> 1) - 1):
            for ow in (px >> 1, (px >> 1) - 1):
                lr, lc = py - 2 * oh, px - 2 * ow
                if 0 <= oh < OH and 0 <= ow < OH and lr <= 2 and lc <= 2 and arg2h[n, ch, oh + 1, ow + 1] == lr * 3 + lc:
                    g += dp2h[n, ch, oh + 1, ow + 1]
        return g

    def x_of(ch):
        return y2h[n, ch, py, px] if 0 <= ch < C2 else 0.0

    def sc_of(ch):
        return BIAS + ALPHA * sum(x_of(k) ** 2 for k in range(ch - 2, ch + 3))

    a = sum(g_of(k) * x_of(k) * sc_of(k) ** (-BETA - 1.0) for k in range(c - 2, c + 3))
    g = g_of(c)
    u = g * sc_of(c) ** -BETA
    r = u - 2.0 * ALPHA * BETA * x_of(c) * a
    return (r if x_of(c) > 0 else 0.0), g, u, [g_of(k) for k in range(c - 4, c + 5)]


for kind in PERT:
    tot = 0
    lanes = np.zeros(64, np.int64)
    phase = np.zeros(10, np.int64)
    shown = 0
    ntr = 0
    for t in range(TRIALS):
        dy2.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            perturb(kind)
        plb()
        torch.cuda.synchronize()
        ntr += 1
        bad = (dy2 != ref)
        k = int(bad.sum())
        if k == 0:
            continue
        tot += k
        idx = torch.nonzero(bad).cpu().numpy()
        now = dy2[bad].cpu().numpy()
        was = ref[bad].cpu().numpy()
        for (n, c, hh, ww), a, b in zip(idx, now, was):
            py, px = hh - 2, ww - 2
            lanes[(py * H2 + px) % 64] += 1
            phase[c % 10] += 1
            if shown < 12:
                r, g, u, gs = host_terms(n, c, py, px)
                print("   n %3d c %3d py %2d px %2d lane %2d | now % .6e ref % .6e | exact % .6e g % .6e u % .6e | g[c-4..c+4] %s"
                      % (n, c, py, px, (py * H2 + px) % 64, a, b, r, g, u, " ".join("%.2e" % v for v in gs)), flush=True)
                shown += 1
    print("perturber %-7s trials %d  differing elements %d" % (kind, ntr, tot), flush=True)
    if tot:
        print("   by lane quarter:", [int(lanes[q * 16:(q + 1) * 16].sum()) for q in range(4)], " by channel %% 10:", phase.tolist(), flush=True)
ops.set_conv_math("f32")
