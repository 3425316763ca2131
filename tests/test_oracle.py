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


# ---- multi-input composition (config 4) and the general pipeline graph: numpy oracle vs torch autograd ------------------------------
def _grad_check(loss_t, pt, grads, rtol=1e-8, atol=1e-11):
    loss_t.backward()
    for k, v in pt.items():
        want = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))
        np.testing.assert_allclose(grads[k], want, rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.parametrize("method,ratio,T", [("avg", 1, 1), ("maximum", 1, 1), ("concat", 1, 1), ("concat", 2, 4), ("concat", 3, 2),
                                            ("ibias", 1, 3), ("ibias", 3, 2)])
def test_tensor_list_fusion_and_grad(method, ratio, T):
    """apply_tensor_list_fusion (tf_util.py:136-179) incl. vec_seq_concat (99-124) and replicate_auxilliary_tensor (182-192) at
    clips-per-video ratios 1, 2, 3; gradients vs autograd (maximum: with exact ties, which tf.reduce_max shares evenly)."""
    rng = np.random.default_rng(11)
    b, dm = 3, 5
    if method in ("avg", "maximum"):
        a, c = rng.standard_normal((b * T, dm)), rng.standard_normal((b * T, dm))
        if method == "maximum":
            c[0, :3] = a[0, :3]                                 # ties
        dims, fpcs, cpvs = [dm, dm], [T, T], [1, 1]
    elif method == "concat" and ratio == 1:
        a, c = rng.standard_normal((b, dm)), rng.standard_normal((b, 4))
        dims, fpcs, cpvs = [dm, 4], [1, 1], [2, 2]
    else:
        da = dm if method == "ibias" else 4
        a, c = rng.standard_normal((b * ratio * T, dm)), rng.standard_normal((b, da))
        dims, fpcs, cpvs = [dm, da], [T, 1], [ratio, 1]
    out, dim, fpc, cpv, cache = O.tensor_list_fusion([a, c], method, dims, fpcs, cpvs)
    at, ct = t(a).requires_grad_(), t(c).requires_grad_()
    ot, dimt, fpct, cpvt = TC.tensor_list_fusion([at, ct], method, dims, fpcs, cpvs)
    assert (dim, fpc, cpv) == (dimt, fpct, cpvt)
    np.testing.assert_allclose(out, ot.detach().numpy(), rtol=1e-12, atol=1e-12)
    d = rng.standard_normal(out.shape)
    ot.backward(t(d))
    ga, gc = O.tensor_list_fusion_grad(cache, d)
    np.testing.assert_allclose(ga, at.grad.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gc, ct.grad.numpy(), rtol=1e-12, atol=1e-12)


def test_replicate_auxilliary_tensor_tiles_the_whole_batch():
    a = np.arange(6.0).reshape(3, 2)
    r = O.replicate_auxilliary_tensor(a, 2)
    np.testing.assert_array_equal(r, np.concatenate([a, a]))          # a0 a1 a2 a0 a1 a2 -- NOT a0 a0 a1 a1 ...
    np.testing.assert_array_equal(r, TC.replicate_aux(t(a), 2).numpy())
    d = np.arange(12.0).reshape(6, 2)
    at = t(a).requires_grad_()
    TC.replicate_aux(at, 2).backward(t(d))
    np.testing.assert_array_equal(O.replicate_auxilliary_tensor_grad(d, 2, 3), at.grad.numpy())


@pytest.mark.parametrize("layers,fusion,state_dim", [(1, "reshape", 6), (2, "avg", 4), (2, "state", 6), (1, "last", None)])
def test_lstm_classifier_with_state_vs_autograd(layers, fusion, state_dim):
    """lstm.forward_pass_sequence (lstm.py:59-99) with the second input as state: input_state_fc present (state_dim != H) and absent
    (== H), c = h = state in every layer (lstm.py:34-42), fusion reshape / avg / last / state (model.py:136-141)."""
    rng = np.random.default_rng(12)
    b, T, D, H, C = 3, 4, 5, 4, 7
    p = O.init_lstm_classifier_params(rng, "dec/", D, H, layers, fusion, C, state_dim=state_dim, well_scaled=True)
    x = rng.standard_normal((b * T, D))
    state = rng.standard_normal((b, state_dim)) if state_dim else None
    logits, cache = O.lstm_classifier_forward(p, "dec/", x, T, layers, fusion, C, state=state)
    pt = {k: t(v).requires_grad_() for k, v in p.items()}
    xt = t(x).requires_grad_()
    st = t(state).requires_grad_() if state is not None else None
    lt = TC.lstm_classifier(pt, "dec/", xt, T, layers, fusion, C, st)
    assert ("dec/input_state_fc_w" in p) == (state_dim is not None and state_dim != H)
    np.testing.assert_allclose(logits, lt.detach().numpy(), rtol=1e-11, atol=1e-11)
    d = rng.standard_normal(logits.shape)
    g, dx, dstate = O.lstm_classifier_backward(p, cache, d)
    lt.backward(t(d))
    for k in p:
        np.testing.assert_allclose(g[k], pt[k].grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=k)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-9, atol=1e-11)
    if state is not None:
        np.testing.assert_allclose(dstate, st.grad.numpy(), rtol=1e-9, atol=1e-11)


def _encdec_case(rng, enc_layers, dec_layers, fusion, enc_h, dec_h, V, shape=(67, 67, 3), Tf=3, Tw=4, b=2, E=5):
    pe = O.init_params(rng, V, "fc6", enc_h, enc_layers, shape, well_scaled=True, fusion="state")
    p = {"enc/" + k: v for k, v in pe.items()}
    p.update(O.init_lstm_classifier_params(rng, "dec/", E, dec_h, dec_layers, fusion, V, state_dim=V, well_scaled=True))
    frames = rng.integers(0, 256, (b * Tf,) + shape).astype(np.float32) - 104.0
    words = rng.standard_normal((b * Tw, E))
    return p, frames, words


ENCDEC_PIPES = lambda el, dl, eh, dh, fusion: [
    ("enc", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier="lstm", lstm_params=[eh, el, "state"])),
    ("dec", dict(input=["aux", "enc"], representation="nop", classifier="lstm", lstm_params=[dh, dl, fusion]))]


@pytest.mark.parametrize("enc_layers,dec_layers,fusion,enc_h,dec_h,V", [(1, 1, "reshape", 6, 8, 9), (2, 2, "avg", 7, 7, 9),
                                                                          (1, 2, "state", 6, 9, 9), (1, 1, "reshape", 6, 9, 9)])
def test_encdec_vs_autograd(enc_layers, dec_layers, fusion, enc_h, dec_h, V):
    """BASELINE config 4's oracle (encdec_forward / encdec_backward: model.py:128-141, lstm.py:34-42,74-77) against torch autograd of
    the same two pipelines, and against the general graph oracle (model_forward) on the same pipelines.  dec_h == V: no input_state_fc."""
    rng = np.random.default_rng(13)
    Tf, Tw = 3, 4
    p, frames, words = _encdec_case(rng, enc_layers, dec_layers, fusion, enc_h, dec_h, V, Tf=Tf, Tw=Tw)
    enc, dec = dict(layer="fc6", layers=enc_layers), dict(layers=dec_layers, fusion=fusion)
    logits, cache = O.encdec_forward(p, frames, words, Tf, Tw, enc, dec, V)
    assert ("dec/input_state_fc_w" in p) == (dec_h != V)
    rows = logits.shape[0]
    lab = rng.integers(0, V, rows)
    loss, dlogits = O.softmax_xent_mean(logits, O.labels_to_one_hot([[l] for l in lab], V))
    grads = O.encdec_backward(p, cache, dlogits, Tf, enc)
    pipes = ENCDEC_PIPES(enc_layers, dec_layers, enc_h, dec_h, fusion)
    ds = {"main": dict(cpv=1, fpc=Tf), "aux": dict(cpv=1, fpc=Tw)}
    lg2, c2 = O.model_forward(p, pipes, ds, {"main": frames, "aux": words}, V)
    np.testing.assert_allclose(lg2, logits, rtol=1e-13, atol=1e-13)
    g2 = O.model_backward(p, c2, dlogits)
    pt = {k: t(v).requires_grad_() for k, v in p.items()}
    lt = TC.model_logits(pt, pipes, ds, {"main": t(frames), "aux": t(words)}, V)
    np.testing.assert_allclose(logits, lt.detach().numpy(), rtol=1e-9, atol=1e-9)
    ce = F.cross_entropy(lt, torch.tensor(lab))
    assert abs(loss - ce.item()) < 1e-10
    _grad_check(ce, pt, grads, rtol=1e-7, atol=1e-10)
    for k in p:
        np.testing.assert_allclose(g2[k], grads[k], rtol=1e-12, atol=1e-14, err_msg=k)
    assert np.abs(grads["enc/dcnn/conv1W"]).max() > 0


def _two_stream(fusion, cls, layer="fc6", third=None):
    """dcnn on main, dcnn on aux (classifier None), a third pipeline fusing them (model.py:41-79)."""
    a = dict(input=["main"], representation="dcnn", frame_encoding_layer=layer, classifier=None)
    b = dict(input=["aux"], representation="dcnn", frame_encoding_layer=layer, classifier=None)
    c = dict(input=["rgb", "flow"], input_fusion=fusion, representation="nop")
    c.update(cls)
    return [("rgb", a), ("flow", b), ("fuse", c)] + (third or [])


def _graph_params(rng, pipes, V, shape, dims):
    """Initial values for every variable model_forward will read (scoped names)."""
    p = {}
    for name, spec in pipes:
        sc = name + "/"
        if spec["representation"] == "dcnn":
            pe = O.init_params(rng, V, spec["frame_encoding_layer"], 4, 1, shape, classifier="none", well_scaled=True)
            p.update({sc + k: v for k, v in pe.items() if k.startswith("dcnn/")})
        for key, (i, o) in dims.get(name, {}).items():
            if key == "lstm":
                H, L, fu = spec["lstm_params"]
                p.update(O.init_lstm_classifier_params(rng, sc, i, H, L, fu, V, state_dim=o, well_scaled=True))
            else:
                p[sc + key + "_w"] = O.truncated_normal(rng, (i, o), np.sqrt(2.0 / i))
                p[sc + key + "_b"] = np.full(o, 0.1, np.float32)
    return p


@pytest.mark.parametrize("fusion", ["avg", "maximum", "concat"])
def test_two_stream_lrcn_graph_vs_autograd(fusion):
    """The two-stream LRCN: three pipelines, two dcnn towers without a classifier fused by input_fusion avg | maximum | concat at
    equal rates into an LSTM classifier (model.py:18-162, tf_util.py:142-149); gradients into both towers vs autograd."""
    rng = np.random.default_rng(14)
    shape, V, T, b, H = (67, 67, 3), 7, 3, 2, 6
    pipes = _two_stream(fusion, dict(classifier="lstm", lstm_params=[H, 1, "avg"]))
    D = 4096 * (2 if fusion == "concat" else 1)
    p = _graph_params(rng, pipes, V, shape, {"fuse": {"lstm": (D, None)}})
    ds = {"main": dict(cpv=1, fpc=T), "aux": dict(cpv=1, fpc=T)}
    feeds = {k: rng.integers(0, 256, (b * T,) + shape).astype(np.float32) - 104.0 for k in ("main", "aux")}
    logits, cache = O.model_forward(p, pipes, ds, feeds, V)
    assert logits.shape == (b, V)
    lab = rng.integers(0, V, b)
    loss, dlogits = O.softmax_xent_mean(logits, O.labels_to_one_hot([[l] for l in lab], V))
    grads = O.model_backward(p, cache, dlogits)
    pt = {k: t(v).requires_grad_() for k, v in p.items()}
    lt = TC.model_logits(pt, pipes, ds, {k: t(v) for k, v in feeds.items()}, V)
    np.testing.assert_allclose(logits, lt.detach().numpy(), rtol=1e-9, atol=1e-9)
    _grad_check(F.cross_entropy(lt, torch.tensor(lab)), pt, grads, rtol=1e-7, atol=1e-10)
    assert np.abs(grads["rgb/dcnn/conv1W"]).max() > 0 and np.abs(grads["flow/dcnn/conv1W"]).max() > 0


def test_graph_with_feature_pipelines_fc_heads_and_fanout_vs_autograd():
    """A five-pipeline graph using the remaining build_pipeline branches: a dcnn feature pipeline with EARLY frame fusion and no
    classifier, consumed by TWO later pipelines (its gradients add up); representation fc; classifier fc with LATE fusion; a
    pipeline whose output nothing consumes (zero gradients); the last pipeline fuses two logits vectors by avg into an fc head."""
    rng = np.random.default_rng(15)
    shape, V, T, b = (67, 67, 3), 6, 3, 2
    pipes = [
        ("feat", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc7", classifier=None, frame_fusion=("early", "avg"))),
        ("a", dict(input=["feat"], representation="fc", fc_output_dim=9, classifier="fc")),
        ("perframe", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc8", classifier="fc", frame_fusion=("late", "last"))),
        ("unused", dict(input=["feat"], representation="fc", fc_output_dim=5, classifier=None)),
        ("out", dict(input=["a", "perframe", "feat"], input_fusion="concat", representation="nop", classifier="fc")),
    ]
    # concat of three at "ratio None" is not the ratio-1 branch (cpv_ratio is None for != 2 inputs): fuse two, then add the third
    pipes[-1] = ("ab", dict(input=["a", "perframe"], input_fusion="maximum", representation="nop", classifier=None))
    pipes.append(("out", dict(input=["ab", "feat"], input_fusion="concat", representation="nop", classifier="fc")))
    p = _graph_params(rng, pipes, V, shape, {"a": {"fc_convert": (4096, 9)}, "unused": {"fc_convert": (4096, 5)},
                                             "out": {"fc_convert": (V + 4096, V)}})
    # pipeline "a": representation fc (4096 -> 9) and classifier fc (9 -> V) both ask for "fc_convert": TF refuses the second
    ds = {"main": dict(cpv=1, fpc=T)}
    feeds = {"main": rng.integers(0, 256, (b * T,) + shape).astype(np.float32) - 104.0}
    with pytest.raises(ValueError, match="already exists"):
        O.model_forward(p, pipes, ds, feeds, V)
    pipes[1] = ("a", dict(input=["feat"], representation="fc", fc_output_dim=V, classifier="fc"))      # classifier fc = identity
    p["a/fc_convert_w"] = O.truncated_normal(rng, (4096, V), np.sqrt(2.0 / 4096))
    p["a/fc_convert_b"] = np.full(V, 0.1, np.float32)
    logits, cache = O.model_forward(p, pipes, ds, feeds, V)
    assert logits.shape == (b, V)
    lab = rng.integers(0, V, b)
    loss, dlogits = O.softmax_xent_mean(logits, O.labels_to_one_hot([[l] for l in lab], V))
    grads = O.model_backward(p, cache, dlogits)
    pt = {k: t(v).requires_grad_() for k, v in p.items()}
    lt = TC.model_logits(pt, pipes, ds, {"main": t(feeds["main"])}, V)
    np.testing.assert_allclose(logits, lt.detach().numpy(), rtol=1e-9, atol=1e-9)
    _grad_check(F.cross_entropy(lt, torch.tensor(lab)), pt, grads, rtol=1e-7, atol=1e-10)
    assert np.abs(grads["unused/fc_convert_w"]).max() == 0 and np.abs(grads["feat/dcnn/fc7W"]).max() > 0
