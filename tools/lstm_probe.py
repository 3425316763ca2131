#!/usr/bin/env python3
"""Times the LSTM cluster launches (forward, backward) in isolation: [clips] x 16 steps, hidden 256.
usage: lstm_probe.py [clips ...]      (profiles/r04_lstm_step_parts.txt: the same probe on builds with parts of a step compiled out)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vltf_amd.ops as ops

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

T, H = 16, 256
dev = "cuda:0"
for b in [int(a) for a in sys.argv[1:]] or [8, 64]:
    torch.manual_seed(0)
    gx = torch.randn(b * T, 4 * H, device=dev) * 0.5
    kh = torch.randn(H, 4 * H, device=dev) * 0.05
    act, dz = torch.zeros(b * T, 4 * H, device=dev), torch.zeros(b * T, 4 * H, device=dev)
    cseq, hseq, hprev = (torch.zeros(b * T, H, device=dev) for _ in range(3))
    dout = torch.randn(b * T, H, device=dev)
    ws = ops.lstm_seq_ws(b, T, H, dev)
    tf = timed(lambda: ops.lstm_seq_fwd(gx, kh, act, cseq, hseq, hprev, b, T, H, ws=ws))
    tb = timed(lambda: ops.lstm_seq_bwd(dout, kh, act, cseq, dz, b, T, H, ws=ws))
    ops.lstm_seq_timed_out(ws)
    print("clips %3d: fwd %.1f us (%.2f per step)  bwd %.1f us (%.2f per step)   [launch included]" % (b, tf, tf / T, tb, tb / T), flush=True)
