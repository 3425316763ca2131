"""Driver of tools/ubench/pk_f32_war.hip (build line in its header): the instruction sequence that failed in pool_lrn_bwd, replayed in
isolation beside synthetic MFMA spinners and beside the library's real kernels (conv3's weight gradient in each conv arithmetic).
usage: pk_f32_war.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vltf_amd import ops

DEV = "cuda:0"
here = os.path.dirname(os.path.abspath(__file__))
ops.fill(torch.zeros(4, device=DEV), 0.0)                 # loads the library (and torch's HIP runtime) first
L = C.CDLL(os.path.join(here, "ubench", "libpkwar.so"))
L.pkwar_victim.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
L.pkwar_spin.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_void_p]

N = 256
conv3 = ops.Conv(256, 13, 13, 384, 3, 3, 1, 1)
conv3.set_halo(1, 1, 1, 1)
p2 = torch.zeros(N, 256, 15, 15, device=DEV)
p2[:, :, 1:-1, 1:-1] = torch.randn(N, 256, 13, 13, device=DEV)
dy3 = torch.zeros(N, 384, 15, 15, device=DEV)
dy3[:, :, 1:-1, 1:-1] = torch.randn(N, 384, 13, 13, device=DEV) * 1e-3
dw3 = torch.zeros(3, 3, 256, 384, device=DEV)
db3 = torch.zeros(384, device=DEV)
wsb = 0
for m in ("f32", "bf16x3", "bf16x6"):
    ops.set_conv_math(m)
    wsb = max(wsb, conv3.wgrad_ws_bytes(N))
ws = torch.zeros(wsb // 4 + 64, device=DEV)
errs = torch.zeros(16, device=DEV, dtype=torch.int32)
sink = torch.zeros(4096, device=DEV)
side = torch.cuda.Stream()
MODES = {0: "v_pk_mul_f32 ; ds_read2_b64 (sources set early)", 1: "rsq/sqrt -> v_pk_mul_f32 ; ds_read2_b64 (the failing sequence)",
         2: "rsq/sqrt -> 2 x v_mul_f32 ; ds_read2_b64", 3: "rsq/sqrt -> v_pk_mul_f32 ; ds_read2_b64 elsewhere (no WAR)"}


def perturb(kind):
    if kind == "none":
        return
    with torch.cuda.stream(side):
        s = torch.cuda.current_stream().cuda_stream
        if kind == "spin bf16 same CU":
            L.pkwar_spin(1, 256, 8192, 20000, sink.data_ptr(), s)
        elif kind == "spin bf16 other CUs":
            L.pkwar_spin(1, 128, 90 * 1024, 20000, sink.data_ptr(), s)
        elif kind == "spin f32 same CU":
            L.pkwar_spin(0, 256, 8192, 10000, sink.data_ptr(), s)
        else:
            ops.set_conv_math(kind.split()[1])
            for _ in range(3):
                conv3.wgrad(p2, dy3, dw3, ws, db3 if conv3.fuses_bias() else None)


for kind in ("none", "spin bf16 same CU", "spin bf16 other CUs", "spin f32 same CU", "wgrad f32", "wgrad bf16x3", "wgrad bf16x6"):
    for mode, gap in ((0, 0), (1, 0), (2, 0), (3, 0), (1, 1), (1, 3), (1, 7)):
        errs.zero_()
        torch.cuda.synchronize()
        perturb(kind)
        other = "other" in kind
        for _ in range(4):
            rc = L.pkwar_victim(mode, gap, 512 if other else 2048, 90 * 1024 if other else 12544, 2000, errs.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        e = errs.cpu().numpy()[:12].reshape(4, 3)
        print("%-20s | %-62s gap %d | wrong results by lane quarter (lo, hi, load): %s%s" %
              (kind, MODES[mode], gap, " | ".join("%d %d %d" % tuple(r) for r in e), "   <== WRONG" if e.sum() else ""), flush=True)
ops.set_conv_math("f32")
