"""torch-CPU restatement of the LRCN train step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The same op graph as oracle/lrcn_oracle.py (same reference citations there) written with torch functional ops + autograd on the
CPU: an independent implementation that (a) pins the numpy oracle in tests/test_oracle.py and (b) is the multi-threaded CPU
baseline bench.py times beside the GPU (``cpu_baseline``, kind "port": the reference's own CPU path is TensorFlow 1.x, which is
not installable here -- SURVEY.md 8c/8d).  PARITY UNPINNED, like the numpy oracle.  Only tests/ and bench.py's cpu_baseline
leg import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import lrcn_oracle as O


def conv_same(x_nhwc, w_hwio, stride, group=1):
    """tf.nn.conv2d(padding='SAME') per group (alexnet.py:15-31), NHWC / HWIO."""
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_hwio.permute(3, 2, 0, 1)
    kh, kw = w.shape[2], w.shape[3]
    _, pt, pb = O.same_pad(x.shape[2], kh, stride)
    _, pl, pr = O.same_pad(x.shape[3], kw, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, stride=stride, groups=group).permute(0, 2, 3, 1)


def lstm(x, kernel, bias):
    """BasicLSTMCell over dynamic_rnn, zero state, gate order i, j, f, o, forget_bias 1 (lstm.py:9-20,102-143)."""
    b, tt, _ = x.shape
    hd = kernel.shape[1] // 4
    h = torch.zeros(b, hd, dtype=x.dtype)
    c = torch.zeros(b, hd, dtype=x.dtype)
    outs = []
    for s in range(tt):
        z = torch.cat([x[:, s], h], 1) @ kernel + bias
        i, j, f, o = z.chunk(4, 1)
        c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h = torch.tanh(c) * torch.sigmoid(o)
        outs.append(h)
    return torch.stack(outs, 1), c, h


def lrcn_logits(p, frames, fpc, final_layer="fc6", lstm_layers=1, fusion="avg", classifier="lstm", frame_fusion=None):
    """p: {tf variable name: tensor}; frames [B*T,H,W,3] NHWC.  Classifier lstm, or fc with early / late frame fusion."""
    a = frames
    for name, kh, kw, co, s, g in O.ALEXNET_CONVS:
        a = torch.relu(conv_same(a, p["dcnn/%sW" % name], s, g) + p["dcnn/%sb" % name])
        if name in ("conv1", "conv2"):     # tf.nn.lrn(radius 2, alpha 2e-5, beta .75, bias 1): torch divides alpha by the window size
            a = F.local_response_norm(a.permute(0, 3, 1, 2), 5, alpha=1e-4, beta=0.75, k=1.0).permute(0, 2, 3, 1)
        if name in ("conv1", "conv2", "conv5"):
            a = F.max_pool2d(a.permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    a = torch.relu(a.reshape(a.shape[0], -1) @ p["dcnn/fc6W"] + p["dcnn/fc6b"])
    if final_layer != "fc6":
        a = torch.relu(a @ p["dcnn/fc7W"] + p["dcnn/fc7b"])
    if final_layer not in ("fc6", "fc7"):
        a = a @ p["dcnn/fc8W"] + p["dcnn/fc8b"]                 # alexnet.py:275: xw_plus_b, no ReLU
    if classifier == "fc":                                      # model.py:103-119,149-151
        fuse = lambda v: v.reshape(-1, fpc, v.shape[1]).mean(1) if frame_fusion[1] == "avg" else v.reshape(-1, fpc, v.shape[1])[:, -1]
        if frame_fusion and frame_fusion[0] == "early" and fpc > 1:
            a = fuse(a)
        if "fc_convert_w" in p:
            a = a @ p["fc_convert_w"] + p["fc_convert_b"]
        if frame_fusion and frame_fusion[0] == "late" and fpc > 1:
            a = fuse(a)
        return a
    x = a.reshape(-1, fpc, a.shape[1])
    for l in range(lstm_layers):
        x, _, _ = lstm(x, p["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
                       p["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l])
    f = x.mean(1) if fusion == "avg" else x[:, -1]              # `state`: final h of the last layer = its output at T-1
    head = "fc_convert" if fusion == "state" else "output_fc"  # model.py:137-141 vs lstm.py:88-90
    return f @ p[head + "_w"] + p[head + "_b"]


def train_step(p, frames, labels, fpc, lr, clip_norm, **kw):
    """One clipped-SGD step in place on the leaf tensors of p (requires_grad).  Returns (loss, global grad norm)."""
    for v in p.values():
        v.grad = None
    loss = F.cross_entropy(lrcn_logits(p, frames, fpc, **kw), labels)
    loss.backward()
    gn = torch.sqrt(sum((v.grad ** 2).sum() for v in p.values()))
    scale = clip_norm / max(float(gn), clip_norm) if clip_norm > 0 else 1.0
    with torch.no_grad():
        for v in p.values():
            v -= lr * scale * v.grad
    return float(loss.detach()), float(gn)
