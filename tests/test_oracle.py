"""Cross-check of the numpy oracle (oracle/lrcn_oracle.py) against torch-CPU functional ops and
autograd (oracle/torch_cpu.py) -- an independent implementation of the same TF-1.x op semantics.
The reference ships no golden vectors ("parity unpinned"), so this is what pins the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lrcn_oracle as O
from oracle import torch_cpu as TC

torch.set_num_threads(4)


def t(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


torch_conv_same = TC.conv_same


@pytest.mark.parametrize("h,w,k,s,g,ci,co", [
    (23, 23, 11, 4, 1, 3, 8),      # conv1-like, SAME with symmetric pad
    (24, 20, 11, 4, 1, 3, 8),      # asymmetric SAME pad (before < after)
    (9, 9, 5, 1, 2, 6, 8),         # grouped 5x5
    (7, 6, 3, 1, 2, 8, 6),         # grouped 3x3
    (7, 7, 3, 1, 1, 4, 6),
])
def test_conv_fwd_bwd(h, w, k, s, g, ci, co):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, h, w, ci)).astype(np.float32)
    wt = rng.standard_normal((k, k, ci // g, co)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    y = O.grouped_conv(x, wt, b, s, g)
    xt, wtt, bt = t(x).requires_grad_(), t(wt).requires_grad_(), t(b).requires_grad_()
    yt = torch_conv_same(xt, wtt, s, g) + bt
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    dy = rng.standard_normal(y.shape)
    yt.backward(t(dy))
    dx, dw, db = O.grouped_conv_grad(x, wt, dy, s, g)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dw, wtt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9, atol=1e-9)


def test_same_pad_rule():
    assert O.same_pad(227, 11, 4) == (57, 4, 4)       # conv1: 57x57, not Caffe's 55x55
    assert O.same_pad(224, 11, 4) == (56, 3, 4)       # asymmetric
    assert O.same_pad(28, 5, 1) == (28, 2, 2)
    assert O.same_pad(13, 3, 1) == (13, 1, 1)
    assert O.valid_out(57, 3, 2) == 28 and O.valid_out(28, 3, 2) == 13 and O.valid_out(13, 3, 2) == 6


def test_lrn_fwd_bwd():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((2, 5, 4, 11)) * 30).astype(np.float32)
    y, _ = O.lrn(x)
    xt = t(x).requires_grad_()
    # torch divides alpha by the window size, TF does not: alpha_torch = alpha_tf * 5
    yt = F.local_response_norm(xt.permute(0, 3, 1, 2), 5, alpha=2e-5 * 5, beta=0.75, k=1.0).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-12)
    dy = rng.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(O.lrn_grad(x, dy), xt.grad.numpy(), rtol=1e-9, atol=1e-11)


def test_maxpool_fwd_bwd_first_argmax():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 9, 11, 3))
    y, arg = O.max_pool_valid(x)
    xt = t(x).requires_grad_()
    yt = F.max_pool2d(xt.permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    np.testing.assert_array_equal(y, yt.detach().numpy())
    dy = rng.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(O.max_pool_valid_grad(x.shape, arg, dy), xt.grad.numpy(), rtol=0, atol=1e-15)
    # ties: all-equal window routes to the FIRST element in scan order
    z = np.zeros((1, 3, 3, 1))
    _, a = O.max_pool_valid(z)
    assert a[0, 0, 0, 0] == 0
    dz = O.max_pool_valid_grad(z.shape, a, np.ones((1, 1, 1, 1)))
    assert dz[0, 0, 0, 0] == 1 and dz.sum() == 1


torch_lstm = TC.lstm


def test_lstm_fwd_bwd():
    rng = np.random.default_rng(3)
    b, tt, d, hd = 3, 5, 7, 4
    x = rng.standard_normal((b, tt, d))
    k = rng.standard_normal((d + hd, 4 * hd)) * 0.5
    bias = rng.standard_normal(4 * hd) * 0.1
    out, (c, h), cache = O.lstm_layer_forward(x, k, bias)
    xt, kt, bt = t(x).requires_grad_(), t(k).requires_grad_(), t(bias).requires_grad_()
    ot, ct, ht = torch_lstm(xt, kt, bt)
    np.testing.assert_allclose(out, ot.detach().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(c, ct.detach().numpy(), rtol=1e-12, atol=1e-12)
    dout = rng.standard_normal(out.shape)
    ot.backward(t(dout))
    dx, dk, db, _, _ = O.lstm_layer_backward(k, cache, dout)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dk, kt.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9, atol=1e-11)


def test_softmax_xent():
    rng = np.random.default_rng(4)
    z = rng.standard_normal((6, 9)) * 4
    lab = rng.integers(0, 9, 6)
    y = O.labels_to_one_hot([[l] for l in lab], 9)
    loss, dz = O.softmax_xent_mean(z, y)
    zt = t(z).requires_grad_()
    lt = F.cross_entropy(zt, torch.tensor(lab))
    lt.backward()
    assert abs(loss - lt.item()) < 1e-12
    np.testing.assert_allclose(dz, zt.grad.numpy(), rtol=1e-10, atol=1e-14)


def torch_lrcn(p, frames, fpc, final_layer, lstm_layers, fusion):
    return TC.lrcn_logits(p, frames, fpc, final_layer, lstm_layers, fusion)


@pytest.mark.parametrize("final_layer,layers,fusion", [("fc6", 1, "avg"), ("fc7", 2, "last"), ("fc6", 2, "state")])
def test_lrcn_train_step_vs_autograd(final_layer, layers, fusion):
    """End-to-end: tiny 67x67 frames, 2 clips x 3 frames; logits, grads, clipped SGD step."""
    rng = np.random.default_rng(5)
    shape = (67, 67, 3)
    ncls, fpc, b = 7, 3, 2
    p = O.init_params(rng, ncls, final_layer, 8, layers, shape, well_scaled=True, fusion=fusion)
    frames = (rng.integers(0, 256, (b * fpc,) + shape).astype(np.float32) - 104.0)
    lab = rng.integers(0, ncls, b)
    onehot = O.labels_to_one_hot([[l] for l in lab], ncls)
    newp, loss, gn, acc, logits, grads = O.lrcn_train_step(p, frames, onehot, fpc, lr=0.01, clip_norm=0.5,
                                                           final_layer=final_layer, lstm_layers=layers, fusion=fusion)
    pt = {k: t(v).requires_grad_() for k, v in p.items()}
    lg = torch_lrcn(pt, t(frames), fpc, final_layer, layers, fusion)
    lt = F.cross_entropy(lg, torch.tensor(lab))
    lt.backward()
    np.testing.assert_allclose(logits, lg.detach().numpy(), rtol=1e-9, atol=1e-9)
    assert abs(loss - lt.item()) < 1e-10
    tn = float(torch.sqrt(sum((v.grad ** 2).sum() for v in pt.values())))
    assert abs(gn - tn) / tn < 1e-9
    scale = 0.5 / max(tn, 0.5)
    for k in p:
        np.testing.assert_allclose(grads[k], pt[k].grad.numpy(), rtol=1e-7, atol=1e-10, err_msg=k)
        want = (pt[k].detach() - 0.01 * scale * pt[k].grad).numpy().astype(np.float32)
        np.testing.assert_allclose(newp[k], want, rtol=1e-6, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("final_layer,ff", [("fc7", ("early", "avg")), ("fc7", ("late", "last")), ("fc8", ("late", "avg")), ("fc6", None)])
def test_fc_classifier_train_step_vs_autograd(final_layer, ff):
    """Config 1 of BASELINE.json (frame-level AlexNet, classifier fc, early / late frame fusion): oracle gradients vs torch autograd."""
    rng = np.random.default_rng(8)
    shape, ncls, fpc, b = (67, 67, 3), 6, 3, 2
    p = O.init_params(rng, ncls, final_layer, 8, 1, shape, classifier="fc", well_scaled=True)
    frames = (rng.integers(0, 256, (b * fpc,) + shape).astype(np.float32) - 104.0)
    rows = b if ff else b * fpc
    lab = rng.integers(0, ncls, rows)
    onehot = O.labels_to_one_hot([[l] for l in lab], ncls)
    newp, loss, gn, acc, logits, grads = O.lrcn_train_step(p, frames, onehot, fpc, lr=0.01, clip_norm=0.5, final_layer=final_layer,
                                                          classifier="fc", frame_fusion=ff)
    pt = {k: t(v).requires_grad_() for k, v in p.items()}
    lg = TC.lrcn_logits(pt, t(frames), fpc, final_layer, classifier="fc", frame_fusion=ff)
    lt = F.cross_entropy(lg, torch.tensor(lab))
    lt.backward()
    np.testing.assert_allclose(logits, lg.detach().numpy(), rtol=1e-9, atol=1e-9)
    assert abs(loss - lt.item()) < 1e-10
    for k in p:
        np.testing.assert_allclose(grads[k], pt[k].grad.numpy(), rtol=1e-7, atol=1e-10, err_msg=k)


def test_lr_table():
    # train.py:50-109: exp and staircase give identical piecewise-constant tables
    a = O.precompute_learning_rates(0.05, ["exp", "interval", 3, 0.5], 5, 2)
    b = O.precompute_learning_rates(0.05, ["staircase", "interval", 3, 0.5], 5, 2)
    assert a == b == [0.05] * 3 + [0.025] * 3 + [0.0125] * 3 + [0.00625]
    c = O.precompute_learning_rates(0.1, ["exp", "drops", 4, 0.1], 5, 2)      # period = ceil(10/4) = 3
    assert len(c) == 10 and c[0] == 0.1 and abs(c[3] - 0.01) < 1e-12 and abs(c[9] - 1e-4) < 1e-12
    d = O.precompute_learning_rates(0.1, ["exp", "interval", 2, 0.5, 3], 4, 2)  # 3-step offset
    assert d[:3] == [0.1] * 3 and d[3:5] == [0.1, 0.1] and abs(d[5] - 0.05) < 1e-12
    assert O.precompute_learning_rates(0.3, None, 3, 2) == [0.3] * 6


def test_imgproc():
    assert O.center_crop_offsets((240, 320, 3), (227, 227, 3)) == (6, 46)
    hs, ws = O.rand_crop_range((240, 320, 3), (227, 227, 3))
    assert hs == list(range(0, 12)) and ws[-1] == 91      # excludes the last two legal offsets
    img = np.arange(4 * 5 * 3, dtype=np.uint8).reshape(4, 5, 3)
    out = O.process_image(img, (2, 3, 3), (1, 1), [1.0, 2.0, 3.0], mirror=True)
    want = (img[1:3, 1:4, :].astype(np.float32) - np.array([1, 2, 3], np.float32))[:, ::-1, :]
    np.testing.assert_array_equal(out, want)
