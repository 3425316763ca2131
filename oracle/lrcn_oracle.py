"""CPU oracle for the LRCN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product path (``video-learning-tf_amd``) never
imports it and fails loudly when its HIP library is missing.

PARITY UNPINNED.  The reference (npit/video-learning-tf) is Python over TensorFlow 1.x graph
ops.  TensorFlow is not vendored, not pinned (reference ``dependencies.txt:1-4``) and not
installable here, and the reference ships no tests, golden vectors or known-answer values
(SURVEY.md section 4, 8c).  This file therefore restates the published TF-1.x op semantics the
reference's call sites rely on, in numpy, and is cross-checked op by op against torch-CPU
functional ops / autograd (an independent implementation) in ``tests/test_oracle.py``.

Layouts follow the reference: activations NHWC, conv kernels HWIO, fc weights [in, out],
LSTM kernel [D+H, 4H] with gate order i, j, f, o.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math

import numpy as np

F64 = np.float64


# ----------------------------------------------------------------------------------------------
# TF padding rule used by every conv of models/alexnet/alexnet.py (padding="SAME") and every
# max-pool (padding="VALID").
# ----------------------------------------------------------------------------------------------
def same_pad(in_size: int, k: int, s: int):
    """TF 'SAME': out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = floor(pad/2)."""
    out = -(-in_size // s)
    pad_total = max((out - 1) * s + k - in_size, 0)
    before = pad_total // 2
    return out, before, pad_total - before


def valid_out(in_size: int, k: int, s: int) -> int:
    return (in_size - k) // s + 1


def _im2col(x, kh, kw, s, pt, pl, oh, ow):
    """x [N,H,W,C] -> cols [N,oh,ow,kh,kw,C] (zero padded)."""
    n, h, w, c = x.shape
    pb = max((oh - 1) * s + kh - h - pt, 0)
    pr = max((ow - 1) * s + kw - w - pl, 0)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = np.empty((n, oh, ow, kh, kw, c), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + (oh - 1) * s + 1:s, j:j + (ow - 1) * s + 1:s, :]
    return cols


def conv2d_same(x, w, stride, dtype=F64):
    """tf.nn.conv2d(x, w, [1,s,s,1], 'SAME'), NHWC x HWIO (alexnet.py:21)."""
    x = x.astype(dtype, copy=False)
    w = w.astype(dtype, copy=False)
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    assert ci == c
    oh, pt, _ = same_pad(h, kh, stride)
    ow, pl, _ = same_pad(wd, kw, stride)
    cols = _im2col(x, kh, kw, stride, pt, pl, oh, ow).reshape(n * oh * ow, kh * kw * c)
    return (cols @ w.reshape(kh * kw * c, co)).reshape(n, oh, ow, co)


def conv2d_same_grad(x, w, dy, stride, dtype=F64, need_dx=True):
    """Gradients of conv2d_same wrt x and w (what tf.gradients yields for alexnet.py:21)."""
    x = x.astype(dtype, copy=False)
    w = w.astype(dtype, copy=False)
    dy = dy.astype(dtype, copy=False)
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    oh, pt, _ = same_pad(h, kh, stride)
    ow, pl, _ = same_pad(wd, kw, stride)
    cols = _im2col(x, kh, kw, stride, pt, pl, oh, ow).reshape(n * oh * ow, kh * kw * c)
    dy2 = dy.reshape(n * oh * ow, co)
    dw = (cols.T @ dy2).reshape(kh, kw, ci, co)
    dx = None
    if need_dx:
        dcols = (dy2 @ w.reshape(kh * kw * c, co).T).reshape(n, oh, ow, kh, kw, c)
        pb = max((oh - 1) * stride + kh - h - pt, 0)
        pr = max((ow - 1) * stride + kw - wd - pl, 0)
        dxp = np.zeros((n, h + pt + pb, wd + pl + pr, c), dtype=dtype)
        for i in range(kh):
            for j in range(kw):
                dxp[:, i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride, :] += dcols[:, :, :, i, j, :]
        dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    return dx, dw


def grouped_conv(x, w, b, stride, group, dtype=F64):
    """dcnn.conv (alexnet.py:15-31): split input on C and kernel on C_out (axis 3), SAME conv
    per group, concat on C, bias_add."""
    if group == 1:
        y = conv2d_same(x, w, stride, dtype)
    else:
        cin = x.shape[-1] // group
        cout = w.shape[-1] // group
        y = np.concatenate([conv2d_same(x[..., g * cin:(g + 1) * cin], w[..., g * cout:(g + 1) * cout], stride, dtype)
                            for g in range(group)], axis=3)
    return y + b.astype(dtype)


def grouped_conv_grad(x, w, dy, stride, group, dtype=F64, need_dx=True):
    db = dy.astype(dtype).sum(axis=(0, 1, 2))
    if group == 1:
        dx, dw = conv2d_same_grad(x, w, dy, stride, dtype, need_dx)
        return dx, dw, db
    cin = x.shape[-1] // group
    cout = w.shape[-1] // group
    dxs, dws = [], []
    for g in range(group):
        dxg, dwg = conv2d_same_grad(x[..., g * cin:(g + 1) * cin], w[..., g * cout:(g + 1) * cout],
                                    dy[..., g * cout:(g + 1) * cout], stride, dtype, need_dx)
        dxs.append(dxg)
        dws.append(dwg)
    dx = np.concatenate(dxs, axis=3) if need_dx else None
    return dx, np.concatenate(dws, axis=3), db


def relu(x):
    return np.maximum(x, 0)


def relu_grad(y, dy):
    """TF ReluGrad: dy where y > 0, else 0 (alexnet.py:77)."""
    return dy * (y > 0)


def lrn(x, radius=2, alpha=2e-5, beta=0.75, bias=1.0, dtype=F64):
    """tf.nn.local_response_normalization (alexnet.py:79-89): out = x / (bias + alpha *
    sum_{c'=max(0,c-r)}^{min(C-1,c+r)} x^2)^beta.  alpha is NOT divided by the window size."""
    x = x.astype(dtype, copy=False)
    c = x.shape[-1]
    sq = x * x
    pad = np.pad(sq, [(0, 0)] * (x.ndim - 1) + [(radius, radius)])
    s = np.zeros_like(x)
    for d in range(2 * radius + 1):
        s += pad[..., d:d + c]
    scale = bias + alpha * s
    return x * scale ** (-beta), scale


def lrn_grad(x, dy, radius=2, alpha=2e-5, beta=0.75, bias=1.0, dtype=F64):
    """d/dx of lrn: dx_i = dy_i s_i^-b - 2ab x_i sum_{c: |c-i|<=r} dy_c x_c s_c^(-b-1)."""
    x = x.astype(dtype, copy=False)
    dy = dy.astype(dtype, copy=False)
    c = x.shape[-1]
    _, scale = lrn(x, radius, alpha, beta, bias, dtype)
    t = dy * x * scale ** (-beta - 1.0)
    pad = np.pad(t, [(0, 0)] * (x.ndim - 1) + [(radius, radius)])
    acc = np.zeros_like(x)
    for d in range(2 * radius + 1):
        acc += pad[..., d:d + c]
    return dy * scale ** (-beta) - 2.0 * alpha * beta * x * acc


def max_pool_valid(x, k=3, s=2):
    """tf.nn.max_pool ksize k, stride s, VALID (alexnet.py:91-98).  Returns (y, argmax) where
    argmax is the window-local index (kh*k+kw) of the FIRST maximum in row-major scan order
    (TF-CPU MaxPoolGrad routes the gradient there)."""
    n, h, w, c = x.shape
    oh, ow = valid_out(h, k, s), valid_out(w, k, s)
    y = np.full((n, oh, ow, c), -np.inf, dtype=x.dtype)
    arg = np.zeros((n, oh, ow, c), dtype=np.int8)
    for i in range(k):
        for j in range(k):
            v = x[:, i:i + (oh - 1) * s + 1:s, j:j + (ow - 1) * s + 1:s, :]
            upd = v > y
            y = np.where(upd, v, y)
            arg = np.where(upd, i * k + j, arg)
    return y, arg


def max_pool_valid_grad(x_shape, arg, dy, k=3, s=2):
    n, h, w, c = x_shape
    oh, ow = dy.shape[1], dy.shape[2]
    dx = np.zeros(x_shape, dtype=dy.dtype)
    for i in range(k):
        for j in range(k):
            dx[:, i:i + (oh - 1) * s + 1:s, j:j + (ow - 1) * s + 1:s, :] += dy * (arg == i * k + j)
    return dx


def xw_plus_b(x, w, b, dtype=F64):
    """tf.nn.xw_plus_b (alexnet.py:275, tf_util.py:56)."""
    return x.astype(dtype) @ w.astype(dtype) + b.astype(dtype)


def sigmoid(x):
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


# ----------------------------------------------------------------------------------------------
# AlexNet (CaffeNet-style) -- models/alexnet/alexnet.py:49-275
# ----------------------------------------------------------------------------------------------
ALEXNET_CONVS = (
    # name, kh, kw, cout, stride, group      (alexnet.py:60-77, 100-118, 141-202)
    ("conv1", 11, 11, 96, 4, 1),
    ("conv2", 5, 5, 256, 1, 2),
    ("conv3", 3, 3, 384, 1, 1),
    ("conv4", 3, 3, 384, 1, 2),
    ("conv5", 3, 3, 256, 1, 2),
)


def alexnet_param_shapes(num_classes, final_layer="fc6", image_shape=(227, 227, 3)):
    """Variable shapes in creation order, names as the TF checkpoint keys (alexnet.py:59,70-74)."""
    shapes = []
    h, w, c = image_shape
    for name, kh, kw, co, s, g in ALEXNET_CONVS:
        shapes.append(("dcnn/%sW" % name, (kh, kw, c // g, co)))
        shapes.append(("dcnn/%sb" % name, (co,)))
        h, _, _ = same_pad(h, kh, s)
        w, _, _ = same_pad(w, kw, s)
        c = co
        if name in ("conv1", "conv2", "conv5"):
            h, w = valid_out(h, 3, 2), valid_out(w, 3, 2)
    shapes.append(("dcnn/fc6W", (h * w * c, 4096)))
    shapes.append(("dcnn/fc6b", (4096,)))
    if final_layer == "fc6":
        return shapes
    shapes.append(("dcnn/fc7W", (4096, 4096)))
    shapes.append(("dcnn/fc7b", (4096,)))
    if final_layer == "fc7":
        return shapes
    shapes.append(("dcnn/fc8W", (4096, num_classes)))
    shapes.append(("dcnn/fc8b", (num_classes,)))
    return shapes


def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: N(0, stddev) with samples beyond 2 stddev re-drawn (alexnet.py:41)."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def init_params(rng, num_classes=101, final_layer="fc6", lstm_hidden=256, lstm_layers=1,
                image_shape=(227, 227, 3), classifier="lstm", stddev=0.05, well_scaled=False, fusion="avg"):
    """Reference initialisers: conv/fc W ~ truncated_normal(0.05), b = 0.1 (alexnet.py:40-46,
    tf_util.py:44-45); BasicLSTMCell kernel glorot-uniform, bias 0 (TF default initialiser).
    ``well_scaled`` replaces sigma by sqrt(2/fan_in) so activations stay O(1)."""
    p = {}
    for name, shp in alexnet_param_shapes(num_classes, final_layer, image_shape):
        if name.endswith("W"):
            fan_in = int(np.prod(shp[:-1]))
            sd = math.sqrt(2.0 / fan_in) if well_scaled else stddev
            p[name] = truncated_normal(rng, shp, sd)
        else:
            p[name] = np.full(shp, 0.1, np.float32)
    dim = {"fc6": 4096, "fc7": 4096}.get(final_layer, num_classes)
    if classifier == "lstm":
        d = dim
        for l in range(lstm_layers):
            lim = math.sqrt(6.0 / (d + lstm_hidden + 4 * lstm_hidden))
            p["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l] = rng.uniform(
                -lim, lim, (d + lstm_hidden, 4 * lstm_hidden)).astype(np.float32)
            p["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l] = np.zeros(4 * lstm_hidden, np.float32)
            d = lstm_hidden
        if lstm_hidden != num_classes:
            # fusion `state` (lstm.py:81-93, model.py:137-141): the logits are the final h of the last layer mapped by
            # convert_dim_fc under its DEFAULT name "fc_convert"; every other fusion maps the pooled outputs with "output_fc"
            name = "fc_convert" if fusion == "state" else "output_fc"
            sd = math.sqrt(2.0 / lstm_hidden) if well_scaled else stddev
            p[name + "_w"] = truncated_normal(rng, (lstm_hidden, num_classes), sd)
            p[name + "_b"] = np.full(num_classes, 0.1, np.float32)
    elif classifier == "fc" and dim != num_classes:
        sd = math.sqrt(2.0 / dim) if well_scaled else stddev
        p["fc_convert_w"] = truncated_normal(rng, (dim, num_classes), sd)
        p["fc_convert_b"] = np.full(num_classes, 0.1, np.float32)
    return p


def bf16_round(a):
    """Round to the nearest bfloat16 (ties to even), returned in a's float type: the operand rounding of the bf16 conv path."""
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    return b.view(np.float32).astype(np.asarray(a).dtype if np.asarray(a).dtype.kind == "f" else np.float32)


def alexnet_forward(p, x, final_layer="fc6", dtype=F64, keep=False, q=None):
    """dcnn.create (alexnet.py:49-275).  x: [N,H,W,3] float (BGR, mean-subtracted), NHWC.
    Returns (output, cache).  final_layer: 'fc6' | 'fc7' | anything else -> fc8 logits.

    q (optional, tests only): operand quantiser of the engine's opt-in bf16 conv path (NetConfig.conv_math = "bf16", BASELINE
    config 5; not a reference feature): q = bf16_round reproduces where that path stores a tensor as packed bf16 -- the input
    frames, the weights of every conv and of fc6, the outputs of conv1..conv4 (conv1 / conv2: also what LRN reads), the pooled
    outputs of pool1 / pool2, pool5's output as fc6's operand -- so that device and oracle differ by fp32 rounding only."""
    cache = {"x": x} if keep else {}
    a = x.astype(dtype, copy=False)
    qq = (lambda v: v) if q is None else (lambda v: q(v).astype(dtype))
    a = qq(a)
    for name, kh, kw, co, s, g in ALEXNET_CONVS:
        z = grouped_conv(a, qq(p["dcnn/%sW" % name]), p["dcnn/%sb" % name], s, g, dtype)
        a_in = a
        a = relu(z)
        if name != "conv5":
            a = qq(a)
        if keep:
            cache[name + "_in"] = a_in
            cache[name] = a
        if name in ("conv1", "conv2"):
            r = a
            a, _ = lrn(a, dtype=dtype)
            if keep:
                cache["lrn" + name[-1] + "_in"] = r
                cache["lrn" + name[-1]] = a
        if name in ("conv1", "conv2", "conv5"):
            pin = a
            a, arg = max_pool_valid(a)
            if name != "conv5":
                a = qq(a)
            if keep:
                cache["pool" + name[-1] + "_in_shape"] = pin.shape
                cache["pool" + name[-1] + "_arg"] = arg
                cache["pool" + name[-1]] = a
    n = a.shape[0]
    flat = qq(a.reshape(n, -1))                # (h, w, c)-major flatten (alexnet.py:228)
    fc6 = relu(xw_plus_b(flat, qq(p["dcnn/fc6W"]), p["dcnn/fc6b"], dtype))
    if keep:
        cache["flat"], cache["fc6"] = flat, fc6
    if final_layer == "fc6":
        return fc6, cache
    fc7 = relu(xw_plus_b(fc6, p["dcnn/fc7W"], p["dcnn/fc7b"], dtype))
    if keep:
        cache["fc7"] = fc7
    if final_layer == "fc7":
        return fc7, cache
    fc8 = xw_plus_b(fc7, p["dcnn/fc8W"], p["dcnn/fc8b"], dtype)
    return fc8, cache


def alexnet_backward(p, cache, dout, final_layer="fc6", dtype=F64, gates=None, q=None):
    """Gradients of alexnet_forward wrt every parameter; dout is d(loss)/d(output).

    gates (optional, tests only): the DISCRETE decisions of the forward pass taken from another evaluation of the same network
    -- {"conv1".."conv5", "fc6", "fc7": boolean ReLU masks (output > 0), "pool1_arg", "pool2_arg", "pool5_arg": arg-max maps} in
    this cache's layouts.  A gradient is a continuous function of the inputs only while those decisions stay fixed; an fp32
    evaluation may take a near-tie the other way than this fp64 one, which moves gradient elements by O(1) without either being
    wrong.  With the other evaluation's decisions substituted, the remaining difference is rounding only."""
    gates = gates or {}
    qq = (lambda v: v) if q is None else (lambda v: q(v).astype(dtype))     # see alexnet_forward: the bf16 path's stored operands

    def rgrad(name, dd):
        return dd * gates[name] if name in gates else relu_grad(cache[name], dd)

    g = {}
    d = dout.astype(dtype)
    if final_layer not in ("fc6", "fc7"):
        g["dcnn/fc8W"] = cache["fc7"].T @ d
        g["dcnn/fc8b"] = d.sum(0)
        d = d @ p["dcnn/fc8W"].astype(dtype).T
    if final_layer != "fc6":
        d = rgrad("fc7", d)
        g["dcnn/fc7W"] = cache["fc6"].T @ d
        g["dcnn/fc7b"] = d.sum(0)
        d = d @ p["dcnn/fc7W"].astype(dtype).T
    d = rgrad("fc6", d)
    g["dcnn/fc6W"] = cache["flat"].T @ qq(d)
    g["dcnn/fc6b"] = d.sum(0)
    d = (qq(d) @ qq(p["dcnn/fc6W"].astype(dtype)).T).reshape(cache["pool5"].shape)
    for name, kh, kw, co, s, grp in reversed(ALEXNET_CONVS):
        i = name[-1]
        if name in ("conv1", "conv2", "conv5"):
            d = max_pool_valid_grad(cache["pool%s_in_shape" % i], gates.get("pool%s_arg" % i, cache["pool%s_arg" % i]), d)
        if name in ("conv1", "conv2"):
            d = lrn_grad(cache["lrn%s_in" % i], d, dtype=dtype)
        d = qq(rgrad(name, d))                 # the bf16 path reads this gradient packed: wgrad, dgrad AND the bias gradient
        dx, dw, db = grouped_conv_grad(cache[name + "_in"], qq(p["dcnn/%sW" % name]), d, s, grp, dtype,
                                       need_dx=(name != "conv1"))
        g["dcnn/%sW" % name] = dw
        g["dcnn/%sb" % name] = db
        d = dx
    return g


# ----------------------------------------------------------------------------------------------
# LSTM -- models/lstm/lstm.py:9-20, 59-143 (BasicLSTMCell + MultiRNNCell + dynamic_rnn)
# ----------------------------------------------------------------------------------------------
FORGET_BIAS = 1.0   # tf.contrib.rnn.BasicLSTMCell default, not overridden at lstm.py:17


def lstm_layer_forward(x, kernel, bias, h0=None, c0=None, dtype=F64, q=None):
    """One BasicLSTMCell unrolled by dynamic_rnn over x [B,T,D] (all rows full length, lstm.py:136).
    gates = concat([x_t, h]) @ kernel + bias; split i, j, f, o;
    c' = c*sigmoid(f + 1.0) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o).
    q (tests only, see alexnet_forward): the engine's bf16 mode rounds both operands of the hoisted input projection x @ kernel[:D];
    the recurrent product h @ kernel[D:] stays fp32."""
    x = x.astype(dtype, copy=False)
    kernel = kernel.astype(dtype, copy=False)
    bias = bias.astype(dtype, copy=False)
    b, t, d = x.shape
    hdim = kernel.shape[1] // 4
    h = np.zeros((b, hdim), dtype) if h0 is None else h0.astype(dtype)
    c = np.zeros((b, hdim), dtype) if c0 is None else c0.astype(dtype)
    hs, cs, gates = [], [], []
    if q is not None:
        x = q(x).astype(dtype)
        kernel = np.concatenate([q(kernel[:d]).astype(dtype), kernel[d:]], axis=0)
    for s in range(t):
        z = np.concatenate([x[:, s, :], h], axis=1) @ kernel + bias
        i, j, f, o = np.split(z, 4, axis=1)
        gi, gj, gf, go = sigmoid(i), np.tanh(j), sigmoid(f + FORGET_BIAS), sigmoid(o)
        c_prev = c
        c = c * gf + gi * gj
        h_prev = h
        h = np.tanh(c) * go
        hs.append(h)
        cs.append(c)
        gates.append((gi, gj, gf, go, c_prev, h_prev))
    out = np.stack(hs, axis=1)
    return out, (c, h), {"x": x, "gates": gates, "cs": cs}


def lstm_layer_backward(kernel, cache, dout, dh_last=None, dc_last=None, dtype=F64, q=None):
    """BPTT through lstm_layer_forward.  dout [B,T,H] = d/d(outputs).
    q (tests only): bf16 operands of the three whole-sequence products x^T dz, h_prev^T dz, dz kernel[:D]^T (the engine's bf16 mode);
    the recurrent dz kernel[D:]^T stays fp32."""
    kernel = kernel.astype(dtype, copy=False)
    x = cache["x"]
    b, t, d = x.shape
    qq = (lambda v: v) if q is None else (lambda v: q(v).astype(dtype))
    kq = np.concatenate([qq(kernel[:d]), kernel[d:]], axis=0) if q is not None else kernel
    hdim = kernel.shape[1] // 4
    dk = np.zeros_like(kernel)
    db = np.zeros(4 * hdim, dtype)
    dx = np.zeros_like(x)
    dh = np.zeros((b, hdim), dtype) if dh_last is None else dh_last.astype(dtype)
    dc = np.zeros((b, hdim), dtype) if dc_last is None else dc_last.astype(dtype)
    for s in reversed(range(t)):
        gi, gj, gf, go, c_prev, h_prev = cache["gates"][s]
        c = cache["cs"][s]
        dh = dh + dout[:, s, :]
        tc = np.tanh(c)
        do = dh * tc
        dc = dc + dh * go * (1.0 - tc * tc)
        di = dc * gj
        dj = dc * gi
        df = dc * c_prev
        dc = dc * gf
        dz = np.concatenate([di * gi * (1 - gi), dj * (1 - gj * gj), df * gf * (1 - gf), do * go * (1 - go)], axis=1)
        xin = np.concatenate([x[:, s, :], qq(h_prev)], axis=1)          # cache["x"] is already the rounded operand
        dk += xin.T @ qq(dz)
        db += dz.sum(0)
        dx[:, s, :] = qq(dz) @ kq[:d].T
        dh = dz @ kq[d:].T
    return dx, dk, db, dh, dc


def temporal_fusion(x, method):
    """apply_temporal_fusion (tf_util.py:4-30): x [B,T,H]."""
    if method == "avg":
        return x.mean(axis=1)
    if method == "last":
        return x[:, -1, :]
    if method == "reshape":
        return x.reshape(-1, x.shape[-1])
    raise ValueError("Undefined frame fusion type : %s" % method)


def temporal_fusion_grad(shape, method, d):
    b, t, h = shape
    if method == "avg":
        return np.repeat(d[:, None, :] / t, t, axis=1)
    if method == "last":
        out = np.zeros(shape, d.dtype)
        out[:, -1, :] = d
        return out
    if method == "reshape":
        return d.reshape(shape)
    raise ValueError(method)


def softmax_xent_mean(logits, onehot, dtype=F64):
    """train.py:120-123: mean_b softmax_cross_entropy_with_logits(logits, labels).
    Returns (loss, dlogits) with TF's registered backprop (softmax - labels)/B."""
    z = logits.astype(dtype)
    y = onehot.astype(dtype)
    zmax = z.max(axis=1, keepdims=True)
    e = np.exp(z - zmax)
    lse = np.log(e.sum(axis=1, keepdims=True)) + zmax
    logp = z - lse
    loss = float((-(y * logp).sum(axis=1)).mean())
    dlogits = (np.exp(logp) - y) / z.shape[0]
    return loss, dlogits


def accuracy(logits, onehot):
    """train.py:142-149 / val.py:199-203."""
    return float(np.mean(np.argmax(logits, 1) == np.argmax(onehot, 1)))


def clip_by_global_norm(grads, clip_norm):
    """tf.clip_by_global_norm (train.py:215): scale = clip / max(global_norm, clip)."""
    gn = math.sqrt(sum(float((g.astype(F64) ** 2).sum()) for g in grads.values()))
    scale = clip_norm / max(gn, clip_norm) if clip_norm else 1.0
    return {k: v * scale for k, v in grads.items()}, gn


def adam_update(p, grads, state, lr, clip_norm=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer(lr) (train.py:205-206, defaults beta1 0.9, beta2 0.999, epsilon 1e-8) applied to the globally clipped
    gradients (train.py:213-217).  TF-1.x Adam: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); m, v exponential averages;
    w -= lr_t * m / (sqrt(v) + eps).  state = {"t": int, "m": {...}, "v": {...}} is updated in place; returns the new parameters."""
    clipped, _ = clip_by_global_norm(grads, clip_norm)
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    out = {}
    for k in p:
        g = clipped[k].astype(F64)
        m = state.setdefault("m", {}).get(k, np.zeros_like(g))
        v = state.setdefault("v", {}).get(k, np.zeros_like(g))
        m = beta1 * m + (1.0 - beta1) * g
        v = beta2 * v + (1.0 - beta2) * g * g
        state["m"][k], state["v"][k] = m, v
        out[k] = (p[k].astype(F64) - lr_t * m / (np.sqrt(v) + eps)).astype(np.float32)
    return out


def labels_to_one_hot(labels, num_classes):
    """utils_.py:160-169."""
    onehots = np.zeros((len(labels), num_classes), np.int32)
    for i, l in enumerate(labels):
        onehots[i][l] = 1
    return onehots


# ----------------------------------------------------------------------------------------------
# LRCN pipeline -- models/model.py:18-155 (dcnn representation -> lstm | fc classifier)
# ----------------------------------------------------------------------------------------------
def lrcn_forward(p, frames, fpc, final_layer="fc6", lstm_layers=1, fusion="avg", classifier="lstm",
                 frame_fusion=None, dtype=F64, keep=False, chunk=32, q=None):
    """frames [B*T,H,W,3] NHWC float, fpc = T.  Returns (logits [B,C], cache).  q: see alexnet_forward (tests only)."""
    n = frames.shape[0]
    feats, caches = [], []
    for s in range(0, n, chunk):
        f, c = alexnet_forward(p, frames[s:s + chunk], final_layer, dtype, keep, q)
        feats.append(f)
        caches.append(c)
    feat = np.concatenate(feats, axis=0)
    cache = {"cnn": caches, "feat": feat, "chunk": chunk}
    dim = feat.shape[1]
    if classifier == "lstm":
        x = feat.reshape(-1, fpc, dim)                         # lstm.py:120
        lcaches = []
        for l in range(lstm_layers):
            x, _, lc = lstm_layer_forward(x, p["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
                                          p["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l], dtype=dtype, q=q)
            lcaches.append(lc)
        # fusion `state`: lstm_state[-1].h (model.py:137-138) = the last layer's output at t = T-1 (dynamic_rnn with full
        # sequence lengths, lstm.py:136), no dropout, then convert_dim_fc("fc_convert") (model.py:140-141)
        fused = temporal_fusion(x, "last" if fusion == "state" else fusion)    # lstm.py:83
        cache.update(lstm=lcaches, seq_shape=x.shape, fused=fused)
        head = "fc_convert" if fusion == "state" else "output_fc"
        if head + "_w" in p:
            logits = xw_plus_b(fused, p[head + "_w"], p[head + "_b"], dtype)   # lstm.py:88-90
        else:
            logits = fused
    else:  # classifier fc (model.py:115-119) with optional early/late frame fusion (model.py:103-106,149-151)
        v = feat
        if frame_fusion and frame_fusion[0] == "early" and fpc > 1:
            v = temporal_fusion(v.reshape(-1, fpc, dim), frame_fusion[1])
        cache["fc_in"] = v
        logits = xw_plus_b(v, p["fc_convert_w"], p["fc_convert_b"], dtype) if "fc_convert_w" in p else v
        cache["pre_late"] = logits
        if frame_fusion and frame_fusion[0] == "late" and fpc > 1:
            logits = temporal_fusion(logits.reshape(-1, fpc, logits.shape[1]), frame_fusion[1])
    return logits, cache


def lrcn_backward(p, cache, dlogits, fpc, final_layer="fc6", lstm_layers=1, fusion="avg", dtype=F64, classifier="lstm",
                  frame_fusion=None, gates=None, q=None):
    """Gradients of lrcn_forward wrt every parameter (classifier lstm, or fc with early / late frame fusion).
    gates: see alexnet_backward; arrays cover ALL frames and are sliced per chunk here."""
    g = {}
    d = dlogits.astype(dtype)
    if classifier == "lstm":
        head = "fc_convert" if fusion == "state" else "output_fc"
        if head + "_w" in p:
            g[head + "_w"] = cache["fused"].T @ d
            g[head + "_b"] = d.sum(0)
            d = d @ p[head + "_w"].astype(dtype).T
        d = temporal_fusion_grad(cache["seq_shape"], "last" if fusion == "state" else fusion, d)
        for l in reversed(range(lstm_layers)):
            kname = "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l
            d, dk, db, _, _ = lstm_layer_backward(p[kname], cache["lstm"][l], d, dtype=dtype, q=q)
            g[kname] = dk
            g[kname[:-6] + "bias"] = db
    else:   # classifier fc (model.py:115-119), late fusion after it (149-151), early fusion before it (103-106)
        if frame_fusion and frame_fusion[0] == "late" and fpc > 1:
            d = temporal_fusion_grad((d.shape[0], fpc, d.shape[1]), frame_fusion[1], d).reshape(-1, d.shape[1])
        if "fc_convert_w" in p:
            g["fc_convert_w"] = cache["fc_in"].T @ d
            g["fc_convert_b"] = d.sum(0)
            d = d @ p["fc_convert_w"].astype(dtype).T
        if frame_fusion and frame_fusion[0] == "early" and fpc > 1:
            d = temporal_fusion_grad((d.shape[0], fpc, d.shape[1]), frame_fusion[1], d)
    dfeat = d.reshape(-1, d.shape[-1])
    chunk = cache["chunk"]
    for ci, cc in enumerate(cache["cnn"]):
        gsl = {k: v[ci * chunk:(ci + 1) * chunk] for k, v in gates.items()} if gates else None
        gc = alexnet_backward(p, cc, dfeat[ci * chunk:(ci + 1) * chunk], final_layer, dtype, gsl, q)
        for k, v in gc.items():
            g[k] = g[k] + v if k in g else v
    return g


def lrcn_train_step(p, frames, onehot, fpc, lr, clip_norm=0.0, final_layer="fc6", lstm_layers=1,
                    fusion="avg", dtype=F64, chunk=32, classifier="lstm", frame_fusion=None):
    """One single-tier SGD step (train.py:199-222): loss, grads, global-norm clip, w -= lr*g.
    Returns (new_params, loss, global_norm, accuracy, logits, grads_unclipped)."""
    logits, cache = lrcn_forward(p, frames, fpc, final_layer, lstm_layers, fusion, classifier, frame_fusion, dtype, True, chunk)
    loss, dlogits = softmax_xent_mean(logits, onehot, dtype)
    grads = lrcn_backward(p, cache, dlogits, fpc, final_layer, lstm_layers, fusion, dtype, classifier, frame_fusion)
    clipped, gn = clip_by_global_norm(grads, clip_norm)
    new_p = {k: (p[k].astype(dtype) - lr * clipped[k]).astype(np.float32) for k in p}
    return new_p, loss, gn, accuracy(logits, onehot), logits, grads


# ----------------------------------------------------------------------------------------------
# Host image processing -- dataset_.py:444-501, 521-530, 571-577
# ----------------------------------------------------------------------------------------------
def center_crop_offsets(raw_shape, want_shape):
    """compute_crop center mode: floor((raw - want)/2) (dataset_.py:572-573)."""
    return tuple(int(np.floor((r - w) / 2)) for r, w in zip(raw_shape[:2], want_shape[:2]))


def rand_crop_range(raw_shape, want_shape):
    """compute_crop rand mode: offsets in range(0, raw - want - 1) (dataset_.py:574-576)."""
    return (list(range(0, raw_shape[0] - want_shape[0] - 1)), list(range(0, raw_shape[1] - want_shape[1] - 1)))


def process_image(img_u8, want_shape, crop_yx=None, mean_bgr=None, mirror=False):
    """process_image (dataset_.py:481-501) for already raw-sized frames: crop -> minus mean
    (float32 HWC, mean_image[0] = blue, dataset_.py:521-530) -> optional mirror of the W axis."""
    img = img_u8
    if crop_yx is not None:
        y, x = crop_yx
        img = img[y:y + want_shape[0], x:x + want_shape[1], :]
    img = img.astype(np.float32)
    if mean_bgr is not None:
        img = img - np.asarray(mean_bgr, np.float32).reshape(1, 1, 3)
    if mirror:
        img = img[:, ::-1, :]
    return img


# ----------------------------------------------------------------------------------------------
# LR schedule -- train.py:50-109
# ----------------------------------------------------------------------------------------------
def precompute_learning_rates(base_lr, decay_params, num_batches, epochs):
    total = num_batches * epochs
    if decay_params is None:
        return [base_lr] * total
    offset = 0 if len(tuple(decay_params)) == 4 else decay_params[-1]
    strategy, scheme, freq, factor = tuple(decay_params[:4])
    staircase = strategy == "staircase"
    period = freq if scheme == "interval" else math.ceil(total / freq)
    lrs, idx = [], 0
    while len(lrs) < total:
        fraction = idx // freq if staircase else idx / freq
        lrs.extend([base_lr * pow(factor, fraction)] * period)
        idx += freq
    lrs = lrs[:total]
    if offset:
        lrs = [base_lr] * offset + lrs[0:-offset]
    return lrs


# ----------------------------------------------------------------------------------------------
# Validation aggregation -- val.py:91-110, 158-203
# ----------------------------------------------------------------------------------------------
def clip_fusion_per_video(clip_logits, cpv_list, method="avg"):
    out, pos = [], 0
    for cpv in cpv_list:
        cur = clip_logits[pos:pos + cpv]
        out.append(cur.mean(axis=0) if method == "avg" else cur[-1])
        pos += cpv
    assert pos == len(clip_logits)
    return np.stack(out).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# Multi-input composition -- tf_util.py:99-124,136-192; models/model.py:61-79,128-141; lstm.py:34-42,59-99
# (the pieces BASELINE config 4, the encoder-decoder description model, is assembled from)
# ----------------------------------------------------------------------------------------------
def replicate_auxilliary_tensor(aux, tile_num):
    """tf_util.py:182-192, literally: reshape to [1, -1], tile [tile_num, 1], reshape to [-1, dim] -- i.e. the WHOLE batch of
    aux vectors repeated tile_num times in sequence (a0..aB-1, a0..aB-1, ...), not each vector repeated in place."""
    if tile_num <= 1:
        return aux
    return np.tile(aux.reshape(1, -1), (tile_num, 1)).reshape(-1, aux.shape[-1])


def replicate_auxilliary_tensor_grad(d, tile_num, rows):
    if tile_num <= 1:
        return d
    return d.reshape(tile_num, rows, d.shape[-1]).sum(axis=0)


def vec_seq_concat(seq, vec, sequence_length, order="vecfirst"):
    """tf_util.py:99-124: every vector of vec [B, dv] is repeated sequence_length times (tile on axis 1, reshape) and
    concatenated column-wise with seq [B*T, ds]."""
    rep = np.tile(vec, (1, sequence_length)).reshape(-1, vec.shape[-1])
    return np.concatenate([rep, seq] if order == "vecfirst" else [seq, rep], axis=1)


def tensor_list_fusion(inputs, method, dims, fpcs, cpvs):
    """apply_tensor_list_fusion (tf_util.py:136-179) -> (tensor, dim, fpc, cpv, cache for the gradient)."""
    cpv_ratio = int(cpvs[0] / cpvs[1]) if len(inputs) == 2 else None
    if method == "avg":
        return np.mean(np.stack(inputs), axis=0), dims[0], fpcs[0], cpvs[0], ("avg", len(inputs))
    if method == "maximum":
        st = np.stack(inputs)
        mx = st.max(axis=0)
        return mx, dims[0], fpcs[0], cpvs[0], ("maximum", st == mx, len(inputs))
    if method == "concat":
        if cpv_ratio == 1:
            return np.concatenate(inputs, axis=1), sum(dims), fpcs[0], cpvs[0], ("concat", list(dims))
        aux = replicate_auxilliary_tensor(inputs[1], cpv_ratio)
        return (vec_seq_concat(inputs[0], aux, fpcs[0]), sum(dims), fpcs[0], cpvs[0],
                ("vecseq", list(dims), cpv_ratio, inputs[1].shape[0], fpcs[0]))
    if method == "ibias":
        main, aux = inputs
        rows_aux = aux.shape[0]
        if cpv_ratio != 1:
            aux = replicate_auxilliary_tensor(aux, cpv_ratio)
        mdim, adim = dims
        combo = np.concatenate([aux.reshape(-1, 1, adim), main.reshape(-1, fpcs[0], mdim)], axis=1)   # needs adim == mdim, as in TF
        return combo.reshape(-1, mdim), mdim, fpcs[0] + 1, cpvs[0], ("ibias", fpcs[0], mdim, cpv_ratio, rows_aux)
    raise ValueError("Unknown fusion method: [%s]" % method)


def tensor_list_fusion_grad(cache, d):
    """-> list of gradients, one per fused input."""
    kind = cache[0]
    if kind == "avg":
        return [d / cache[1]] * cache[1]
    if kind == "maximum":
        # tf.reduce_max's registered gradient (_MinOrMaxGrad): the inputs equal to the maximum share it evenly
        hit = cache[1]
        return [d * hit[i] / hit.sum(axis=0) for i in range(cache[2])]
    if kind == "concat":
        out, pos = [], 0
        for dim in cache[1]:
            out.append(d[:, pos:pos + dim])
            pos += dim
        return out
    if kind == "vecseq":
        (dmain, daux), ratio, rows_aux, T = cache[1], cache[2], cache[3], cache[4]
        dvec = d[:, :daux].reshape(-1, T, daux).sum(axis=1)            # vecfirst: the tiled aux occupies the first columns
        return [d[:, daux:], replicate_auxilliary_tensor_grad(dvec, ratio, rows_aux)]
    if kind == "ibias":
        T, mdim, ratio, rows_aux = cache[1], cache[2], cache[3], cache[4]
        d3 = d.reshape(-1, T + 1, mdim)
        return [d3[:, 1:, :].reshape(-1, mdim), replicate_auxilliary_tensor_grad(d3[:, 0, :], ratio, rows_aux)]
    raise ValueError(kind)


def lstm_classifier_forward(p, scope, x, T, layers, fusion, out_dim, state=None, dtype=F64):
    """LSTM.build -> lstm.forward_pass_sequence (vectorizer.py:63-69, lstm.py:59-99) followed by the model-level handling of the
    `state` fusion (model.py:136-141).  x [B*T, D]; state (optional) [B, S]: mapped by convert_dim_fc "input_state_fc" when
    S != H (lstm.py:74-77) and used as BOTH c and h of EVERY layer (get_state_tuple, lstm.py:34-42).
    fusion avg | last: pooled outputs -> [dropout] -> output_fc -> [B, out_dim];  reshape: every step's output -> output_fc ->
    [B*T, out_dim];  state: final h of the last layer -> convert_dim_fc "fc_convert" -> [B, out_dim].
    Parameter names are `scope` + the TF names.  Returns (logits, cache)."""
    b = x.shape[0] // T
    hdim = p[scope + "rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel"].shape[1] // 4
    cache = {"T": T, "fusion": fusion, "layers": layers, "scope": scope}
    s = None
    if state is not None:
        s = state.astype(dtype)
        cache["state_in"] = s
        if scope + "input_state_fc_w" in p:
            s = xw_plus_b(s, p[scope + "input_state_fc_w"], p[scope + "input_state_fc_b"], dtype)
    seq = x.astype(dtype).reshape(b, T, -1)
    lc = []
    for l in range(layers):
        pre = scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
        seq, _, c = lstm_layer_forward(seq, p[pre + "kernel"], p[pre + "bias"], h0=s, c0=s, dtype=dtype)
        lc.append(c)
    cache.update(lstm=lc, seq_shape=seq.shape, has_state=s is not None)
    fused = temporal_fusion(seq, "last" if fusion == "state" else fusion)
    cache["fused"] = fused
    head = scope + ("fc_convert" if fusion == "state" else "output_fc")
    cache["head"] = head
    logits = xw_plus_b(fused, p[head + "_w"], p[head + "_b"], dtype) if head + "_w" in p else fused
    return logits, cache


def lstm_classifier_backward(p, cache, dlogits, dtype=F64):
    """-> (gradients of the scope's parameters, d/dx [B*T, D], d/dstate [B, S] or None)."""
    g, scope, head = {}, cache["scope"], cache["head"]
    d = dlogits.astype(dtype)
    if head + "_w" in p:
        g[head + "_w"] = cache["fused"].T @ d
        g[head + "_b"] = d.sum(0)
        d = d @ p[head + "_w"].astype(dtype).T
    d = temporal_fusion_grad(cache["seq_shape"], "last" if cache["fusion"] == "state" else cache["fusion"], d)
    ds = None
    for l in reversed(range(cache["layers"])):
        pre = scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
        d, dk, db, dh0, dc0 = lstm_layer_backward(p[pre + "kernel"], cache["lstm"][l], d, dtype=dtype)
        g[pre + "kernel"], g[pre + "bias"] = dk, db
        if cache["has_state"]:
            ds = dh0 + dc0 if ds is None else ds + dh0 + dc0          # c = h = state in every layer
    dstate = None
    if cache["has_state"]:
        if scope + "input_state_fc_w" in p:
            g[scope + "input_state_fc_w"] = cache["state_in"].T @ ds
            g[scope + "input_state_fc_b"] = ds.sum(0)
            ds = ds @ p[scope + "input_state_fc_w"].astype(dtype).T
        dstate = ds
    return g, d.reshape(-1, d.shape[-1]), dstate


def init_lstm_classifier_params(rng, scope, in_dim, hidden, layers, fusion, out_dim, state_dim=None, stddev=0.05, well_scaled=False):
    """Initialisers of the variables lstm_classifier_forward reads (BasicLSTMCell glorot-uniform kernels / zero biases;
    convert_dim_fc truncated_normal(0.05) / 0.1, tf_util.py:44-45)."""
    p, d = {}, in_dim
    for l in range(layers):
        lim = math.sqrt(6.0 / (d + hidden + 4 * hidden))
        p[scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l] = rng.uniform(-lim, lim, (d + hidden, 4 * hidden)).astype(np.float32)
        p[scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l] = np.zeros(4 * hidden, np.float32)
        d = hidden

    def fc(name, i, o):
        p[name + "_w"] = truncated_normal(rng, (i, o), math.sqrt(2.0 / i) if well_scaled else stddev)
        p[name + "_b"] = np.full(o, 0.1, np.float32)
    if state_dim is not None and state_dim != hidden:
        fc(scope + "input_state_fc", state_dim, hidden)
    if hidden != out_dim:
        fc(scope + ("fc_convert" if fusion == "state" else "output_fc"), hidden, out_dim)
    return p


def encdec_forward(p, frames, words, fpc_frames, fpc_words, enc, dec, num_classes, dtype=F64, chunk=32):
    """BASELINE config 4 as the reference's Model would assemble it from two pipelines (model.py:18-162):
      enc: {input: frames dataset, representation dcnn @ enc["layer"], classifier lstm, lstm_params [H, L, state]}
           -> final h of the last layer -> convert_dim_fc to num_classes when H != num_classes (model.py:136-141)
      dec: {input: [word-vector dataset, enc], representation nop, classifier lstm, lstm_params [H, L, fusion]}
           -> enc's output is the decoder's state vector (model.py:128-134): replicate (cpv ratio 1 here), input_state_fc,
              c = h = state; the word vectors [B*Tw, E] are the sequence; fusion reshape gives per-step logits [B*Tw, classes].
    Variables are scoped by pipeline name ("enc/", "dec/") -- in the reference both LSTMs would claim the same TF names.
    Returns (logits, cache)."""
    pe = {k[len("enc/"):]: v for k, v in p.items() if k.startswith("enc/")}
    enc_out, ec = lrcn_forward(pe, frames, fpc_frames, enc["layer"], enc["layers"], "state", "lstm", None, dtype, True, chunk)
    logits, dc = lstm_classifier_forward(p, "dec/", words, fpc_words, dec["layers"], dec["fusion"], num_classes, state=enc_out, dtype=dtype)
    return logits, {"enc": ec, "dec": dc, "enc_out": enc_out}


def encdec_backward(p, cache, dlogits, fpc_frames, enc, dtype=F64):
    pe = {k[len("enc/"):]: v for k, v in p.items() if k.startswith("enc/")}
    g, _, dstate = lstm_classifier_backward(p, cache["dec"], dlogits, dtype)
    ge = lrcn_backward(pe, cache["enc"], dstate, fpc_frames, enc["layer"], enc["layers"], "state", dtype, "lstm", None)
    g.update({"enc/" + k: v for k, v in ge.items()})
    return g


# ----------------------------------------------------------------------------------------------
# The general pipeline graph -- Model.__init__ / Model.build_pipeline (models/model.py:18-162)
# ----------------------------------------------------------------------------------------------
def convert_dim_fc(p, name, x, out_dim, dtype=F64):
    """tf_util.py:32-60: identity when the width already matches, else x @ <name>_w + <name>_b."""
    if x.shape[-1] == out_dim:
        return x, False
    return xw_plus_b(x, p[name + "_w"], p[name + "_b"], dtype), True


def _sub(p, scope):
    return {k[len(scope):]: v for k, v in p.items() if k.startswith(scope)} if scope else p


def pipeline_forward(p, scope, spec, ins, num_classes, dtype=F64, chunk=32):
    """Model.build_pipeline (model.py:18-155) for one pipeline.
    spec: the YAML keys of settings_.py:167-208 -- input, input_fusion, representation (dcnn | nop | fc), frame_encoding_layer,
    fc_output_dim, classifier (None | fc | lstm), lstm_params [hidden, layers, fusion], frame_fusion (type, method).
    ins: [(tensor, cpv, fpc)] per input, in order.  Variables are named `scope` + the TF names.
    -> (output, cpv, fpc of the output, cache)."""
    tensors = [t for t, _, _ in ins]
    cpvs, fpcs = [c for _, c, _ in ins], [f for _, _, f in ins]
    dims = [int(t.shape[-1]) for t in tensors]
    cpv = cpvs[-1]                        # model.py:44-59: `cpv` is the loop variable, i.e. the LAST input's, whatever the fusion does
    c = {"spec": spec, "scope": scope}
    ftype, fmethod = spec.get("frame_fusion") or (None, None)
    classif = spec.get("classifier")
    if classif is None and ftype == "late":
        raise ValueError("Specified late fusion with no classifier selected")                          # model.py:36-37
    if spec.get("input_fusion") is not None:                                                           # model.py:69-73
        x, dim, fpc, cpv0, fcache = tensor_list_fusion([t.astype(dtype) for t in tensors], spec["input_fusion"], dims, fpcs, cpvs)
        tensors, dims, fpcs, cpvs = [x], [dim], [fpc], [cpv0]
        c["in_fusion"] = fcache
    x, fpc = tensors[0], fpcs[0]
    out_fpc = fpc
    rep = spec["representation"]
    if rep == "dcnn":                                                                                  # model.py:81-84
        layer = spec["frame_encoding_layer"]
        pd = _sub(p, scope)
        feats, cc = [], []
        for s0 in range(0, x.shape[0], chunk):
            f, ch = alexnet_forward(pd, x[s0:s0 + chunk], layer, dtype, True)
            feats.append(f)
            cc.append(ch)
        feat = np.concatenate(feats, axis=0)
        c.update(cnn=cc, chunk=chunk, layer=layer)
    elif rep == "nop":
        feat = x.astype(dtype)
    elif rep == "fc":                                                                                  # vectorizer.py:77-78
        c["rep_in"] = x.astype(dtype)
        feat, c["rep_fc"] = convert_dim_fc(p, scope + "fc_convert", c["rep_in"], spec["fc_output_dim"], dtype)
    else:
        raise ValueError("Undefined representation [%s]" % rep)
    dim = feat.shape[-1]
    c["feat_shape"] = feat.shape
    if ftype == "early" and fpc > 1:                                                                   # model.py:103-106
        feat = temporal_fusion(feat.reshape(-1, fpc, dim), fmethod)
        c["early"] = (fpc, fmethod)
        out_fpc = 1
    if classif is None:                                                                                # model.py:110-112
        return feat, cpv, out_fpc, c
    if classif == "fc":                                                                                # model.py:115-119
        if rep == "fc" and c["rep_fc"] and dim != num_classes:
            raise ValueError("Variable %sfc_convert_w already exists" % scope)                         # tf.get_variable, tf_util.py:48
        c["cls_in"] = feat
        logits, c["cls_fc"] = convert_dim_fc(p, scope + "fc_convert", feat, num_classes, dtype)
    elif classif == "lstm":                                                                            # model.py:120-141
        if fpc == 1:
            raise ValueError("The LSTM classifier requires an fpc greater than 1")
        if ftype not in (None, "none"):
            raise ValueError("The LSTM classifier should be used only with [none] fusion, but it's [%s]" % ftype)
        hidden, layers, lfusion = spec["lstm_params"][:3]
        state = None
        if len(tensors) > 1:                                                                           # the 2nd input is the state vector
            if len(tensors) != 2:
                raise ValueError("too many values to unpack (expected 2)")                             # tf_util.py:184
            ratio = int(cpvs[0] / cpvs[1])
            state = replicate_auxilliary_tensor(tensors[1].astype(dtype), ratio)
            c["state_rep"] = (ratio, tensors[1].shape[0])
        logits, c["lstm"] = lstm_classifier_forward(p, scope, feat, fpc, layers, lfusion, num_classes, state=state, dtype=dtype)
    else:
        raise ValueError("Undefined classifier [%s]" % classif)
    if ftype == "late" and fpc > 1:                                                                    # model.py:149-151
        c["late"] = (logits.shape, fpc, fmethod)
        logits = temporal_fusion(logits.reshape(-1, fpc, num_classes), fmethod)
    return logits, cpv, 1, c


def pipeline_backward(p, c, dout, num_inputs, dtype=F64):
    """-> (gradients of the pipeline's variables, [d/d(input i)], None for an input no gradient flows to: raw frames)."""
    spec, scope = c["spec"], c["scope"]
    g = {}
    d = dout.astype(dtype)
    dstate = None
    classif = spec.get("classifier")
    if "late" in c:
        shape, fpc, fmethod = c["late"]
        d = temporal_fusion_grad((shape[0] // fpc, fpc, shape[1]), fmethod, d).reshape(shape)
    if classif == "fc":
        if c["cls_fc"]:
            g[scope + "fc_convert_w"] = c["cls_in"].T @ d
            g[scope + "fc_convert_b"] = d.sum(0)
            d = d @ p[scope + "fc_convert_w"].astype(dtype).T
    elif classif == "lstm":
        gl, d, dstate = lstm_classifier_backward(p, c["lstm"], d, dtype)
        g.update(gl)
        if dstate is not None:
            ratio, rows = c["state_rep"]
            dstate = replicate_auxilliary_tensor_grad(dstate, ratio, rows)
    if "early" in c:
        fpc, fmethod = c["early"]
        fs = c["feat_shape"]
        d = temporal_fusion_grad((fs[0] // fpc, fpc, fs[1]), fmethod, d).reshape(fs)
    rep = spec["representation"]
    if rep == "dcnn":
        pd = _sub(p, scope)
        for ci, cc in enumerate(c["cnn"]):
            gc = alexnet_backward(pd, cc, d[ci * c["chunk"]:(ci + 1) * c["chunk"]], c["layer"], dtype)
            for k, v in gc.items():
                g[scope + k] = g[scope + k] + v if scope + k in g else v
        d = None                                            # the input is a dataset placeholder
    elif rep == "fc" and c["rep_fc"]:
        g[scope + "fc_convert_w"] = c["rep_in"].T @ d
        g[scope + "fc_convert_b"] = d.sum(0)
        d = d @ p[scope + "fc_convert_w"].astype(dtype).T
    if "in_fusion" in c:
        return g, (tensor_list_fusion_grad(c["in_fusion"], d) if d is not None else [None] * num_inputs)
    return g, [d] + ([dstate] if num_inputs > 1 else [])


def model_forward(p, pipelines, datasets, feeds, num_classes, dtype=F64, chunk=32):
    """Model.__init__ (model.py:157-162): the pipelines are built in order, an input is a dataset tag (a placeholder: feeds[tag],
    with datasets[tag] = {"cpv", "fpc"}) or the output of an earlier pipeline; the LAST pipeline's output is the logits.
    pipelines: [(name, spec)].  With more than one pipeline every variable is scoped "<pipeline>/<tf name>" (in the reference a
    second dcnn gets TF's automatic "dcnn_1/" name scope and a second LSTM / fc_convert cannot be created at all).
    -> (logits, cache)."""
    scoped = len(pipelines) > 1
    # sess.run(logits) evaluates only what the last pipeline depends on; the others exist as variables and are never run
    by_name, needed, stack = dict(pipelines), set(), [pipelines[-1][0]]
    while stack:
        n = stack.pop()
        if n not in needed:
            needed.add(n)
            stack += [i for i in by_name[n]["input"] if i in by_name]
    outs, shapes, cache = {}, {}, {"order": [n for n, _ in pipelines if n in needed], "pipes": {}, "sources": {}}
    for name, spec in pipelines:
        if name not in needed:
            continue
        ins = []
        for src in spec["input"]:
            if src in outs:
                ins.append((outs[src],) + shapes[src])
            else:
                ins.append((feeds[src], datasets[src]["cpv"], datasets[src]["fpc"]))
        out, cpv, fpc, c = pipeline_forward(p, name + "/" if scoped else "", spec, ins, num_classes, dtype, chunk)
        outs[name], shapes[name] = out, (cpv, fpc)
        cache["pipes"][name], cache["sources"][name] = c, list(spec["input"])
    cache["outputs"] = outs
    return outs[pipelines[-1][0]], cache


def model_backward(p, cache, dlogits, dtype=F64):
    """Gradients of model_forward's logits wrt every variable in p (zeros for a pipeline nothing downstream consumes)."""
    order = cache["order"]
    dout = {order[-1]: dlogits.astype(dtype)}
    g = {}
    for name in reversed(order):
        if name not in dout:
            continue
        gp, dins = pipeline_backward(p, cache["pipes"][name], dout[name], len(cache["sources"][name]), dtype)
        g.update(gp)
        for src, d in zip(cache["sources"][name], dins):
            if src in cache["pipes"] and d is not None:
                dout[src] = dout[src] + d if src in dout else d
    for k, v in p.items():
        if k not in g:
            g[k] = np.zeros(v.shape, dtype)
    return g


# ----------------------------------------------------------------------------------------------
# imresize -- dataset_.py:481-495, serialize.py:424-425: scipy.misc.imresize(image, shape) = PIL's Image.resize((w, h), BILINEAR) on
# the uint8 array (scipy < 1.2: toimage(arr).resize(size, resample=2)).  Restated from Pillow's published algorithm
# (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) and pinned
# bit for bit against the installed Pillow in tests/test_resize.py.
# ----------------------------------------------------------------------------------------------
RESIZE_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1) over the full axis (box = 0 .. in_size).
    -> (bounds int32 [out, 2] = (first input index, tap count), coefficients int32 [out, ksize] in 2^-22 units, ksize)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # in0, in1 are C floats
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0.0 else v
            wt = 1.0 - v if v < 1.0 else 0.0
            w.append(wt)
            ww += wt
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << RESIZE_PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << RESIZE_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis(img, out_size, axis):
    """One 8-bit pass: out = clip8((2^21 + sum_k pixel * coeff) >> 22) along `axis` of a uint8 array."""
    bounds, kk, ksize = pil_bilinear_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (RESIZE_PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> RESIZE_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def imresize_bilinear_u8(img, out_hw):
    """scipy.misc.imresize(img, (h, w, c)) for a uint8 [H, W, C] image: Pillow's two-pass resample, horizontal pass first (into a
    uint8 intermediate), then vertical; an axis whose size does not change is skipped (ImagingResample: need_horizontal /
    need_vertical), so an image already at the target size comes back unchanged."""
    out = img
    if out_hw[1] != img.shape[1]:
        out = _resample_axis(out, out_hw[1], 1)
    if out_hw[0] != img.shape[0]:
        out = _resample_axis(out, out_hw[0], 0)
    return out
