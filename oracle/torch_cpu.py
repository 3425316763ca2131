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


def lstm(x, kernel, bias, state=None):
    """BasicLSTMCell over dynamic_rnn, gate order i, j, f, o, forget_bias 1 (lstm.py:9-20,102-143); zero state, or
    LSTMStateTuple(state, state) -- c = h = state (get_state_tuple, lstm.py:34-42)."""
    b, tt, _ = x.shape
    hd = kernel.shape[1] // 4
    h = torch.zeros(b, hd, dtype=x.dtype) if state is None else state
    c = torch.zeros(b, hd, dtype=x.dtype) if state is None else state
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


# ---- the multi-input pieces and the general pipeline graph (tf_util.py:99-192, models/model.py:18-162), written with torch ops from
# the TF calls the reference makes; autograd supplies every gradient the numpy oracle derives by hand ---------------------------------
def replicate_aux(aux, tile_num):
    """tf_util.py:182-192: reshape [1, -1] -> tile [tile_num, 1] -> reshape [-1, dim]."""
    if tile_num <= 1:
        return aux
    return aux.reshape(1, -1).repeat(tile_num, 1).reshape(-1, aux.shape[-1])


def vec_seq_concat(seq, vec, length):
    """tf_util.py:99-124, order 'vecfirst': tile on axis 1, one vector per row again, concat in front of the sequence rows."""
    return torch.cat([vec.repeat(1, length).reshape(-1, vec.shape[-1]), seq], dim=1)


def tensor_list_fusion(inputs, method, dims, fpcs, cpvs):
    """apply_tensor_list_fusion (tf_util.py:136-179) -> (tensor, dim, fpc, cpv)."""
    ratio = int(cpvs[0] / cpvs[1]) if len(inputs) == 2 else None
    if method == "avg":
        return torch.stack(inputs).mean(0), dims[0], fpcs[0], cpvs[0]
    if method == "maximum":
        return torch.stack(inputs).amax(0), dims[0], fpcs[0], cpvs[0]        # amax: ties share the gradient evenly, like tf.reduce_max
    if method == "concat":
        if ratio == 1:
            return torch.cat(inputs, dim=1), sum(dims), fpcs[0], cpvs[0]
        return vec_seq_concat(inputs[0], replicate_aux(inputs[1], ratio), fpcs[0]), sum(dims), fpcs[0], cpvs[0]
    if method == "ibias":
        main, aux = inputs
        if ratio != 1:
            aux = replicate_aux(aux, ratio)
        combo = torch.cat([aux.reshape(-1, 1, dims[1]), main.reshape(-1, fpcs[0], dims[0])], dim=1)
        return combo.reshape(-1, dims[0]), dims[0], fpcs[0] + 1, cpvs[0]
    raise ValueError(method)


def _fc(p, name, x, out_dim):
    """convert_dim_fc (tf_util.py:32-60)."""
    return x if x.shape[-1] == out_dim else x @ p[name + "_w"] + p[name + "_b"]


def _fuse_time(v, fpc, method):
    """aggregate_clip_vectors (tf_util.py:126-133)."""
    v3 = v.reshape(-1, fpc, v.shape[-1])
    return {"avg": lambda: v3.mean(1), "last": lambda: v3[:, -1], "reshape": lambda: v3.reshape(-1, v.shape[-1])}[method]()


def dcnn_features(p, scope, frames, final_layer):
    a = frames
    for name, kh, kw, co, s, g in O.ALEXNET_CONVS:
        a = torch.relu(conv_same(a, p[scope + "dcnn/%sW" % name], s, g) + p[scope + "dcnn/%sb" % name])
        if name in ("conv1", "conv2"):
            a = F.local_response_norm(a.permute(0, 3, 1, 2), 5, alpha=1e-4, beta=0.75, k=1.0).permute(0, 2, 3, 1)
        if name in ("conv1", "conv2", "conv5"):
            a = F.max_pool2d(a.permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    a = torch.relu(a.reshape(a.shape[0], -1) @ p[scope + "dcnn/fc6W"] + p[scope + "dcnn/fc6b"])
    if final_layer != "fc6":
        a = torch.relu(a @ p[scope + "dcnn/fc7W"] + p[scope + "dcnn/fc7b"])
    if final_layer not in ("fc6", "fc7"):
        a = a @ p[scope + "dcnn/fc8W"] + p[scope + "dcnn/fc8b"]
    return a


def lstm_classifier(p, scope, x, T, layers, fusion, out_dim, state=None):
    """lstm.forward_pass_sequence (lstm.py:59-99) + the `state` handling of model.py:136-141."""
    hidden = p[scope + "rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel"].shape[1] // 4
    if state is not None:
        state = _fc(p, scope + "input_state_fc", state, hidden)
    seq = x.reshape(-1, T, x.shape[-1])
    h = None
    for l in range(layers):
        pre = scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
        seq, _, h = lstm(seq, p[pre + "kernel"], p[pre + "bias"], state)
    if fusion == "state":
        return _fc(p, scope + "fc_convert", h, out_dim)               # lstm_state[-1].h -> convert_dim_fc (default name)
    return _fc(p, scope + "output_fc", _fuse_time(seq.reshape(-1, hidden), T, fusion), out_dim)


def model_logits(p, pipelines, datasets, feeds, num_classes):
    """Model.__init__ / build_pipeline (model.py:18-162); arguments as oracle.lrcn_oracle.model_forward, tensors in feeds / p."""
    scoped = len(pipelines) > 1
    outs, shapes = {}, {}
    for name, spec in pipelines:
        scope = name + "/" if scoped else ""
        tensors, cpvs, fpcs = [], [], []
        for src in spec["input"]:
            t, (cpv, fpc) = (outs[src], shapes[src]) if src in outs else (feeds[src], (datasets[src]["cpv"], datasets[src]["fpc"]))
            tensors.append(t); cpvs.append(cpv); fpcs.append(fpc)
        dims = [t.shape[-1] for t in tensors]
        cpv = cpvs[-1]
        ftype, fmethod = spec.get("frame_fusion") or (None, None)
        if spec.get("input_fusion") is not None:
            x, dim, fpc, cpv0 = tensor_list_fusion(tensors, spec["input_fusion"], dims, fpcs, cpvs)
            tensors, dims, fpcs, cpvs = [x], [dim], [fpc], [cpv0]
        x, fpc = tensors[0], fpcs[0]
        out_fpc = fpc
        rep = spec["representation"]
        if rep == "dcnn":
            v = dcnn_features(p, scope, x, spec["frame_encoding_layer"])
        elif rep == "fc":
            v = _fc(p, scope + "fc_convert", x, spec["fc_output_dim"])
        else:
            v = x
        if ftype == "early" and fpc > 1:
            v, out_fpc = _fuse_time(v, fpc, fmethod), 1
        cls = spec.get("classifier")
        if cls is None:
            outs[name], shapes[name] = v, (cpv, out_fpc)
            continue
        if cls == "fc":
            v = _fc(p, scope + "fc_convert", v, num_classes)
        else:
            hidden, layers, lfusion = spec["lstm_params"][:3]
            state = replicate_aux(tensors[1], int(cpvs[0] / cpvs[1])) if len(tensors) > 1 else None
            v = lstm_classifier(p, scope, v, fpc, layers, lfusion, num_classes, state)
        if ftype == "late" and fpc > 1:
            v = _fuse_time(v, fpc, fmethod)
        outs[name], shapes[name] = v, (cpv, 1)
    return outs[pipelines[-1][0]]
