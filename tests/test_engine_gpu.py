"""End-to-end parity of the two executor calls (forward logits, one clipped-SGD train step)
against the CPU oracle on identical inputs.  Tolerance: north_star's 1e-3 on logits with
well-scaled weights; parameters after the step to 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import lrcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def make(cfg_kw, shape, b, seed=0, max_clips=None):
    from vltf_amd.engine import LRCNEngine, NetConfig
    cfg = NetConfig(image_shape=shape, **cfg_kw)
    eng = LRCNEngine(cfg, max_clips=max_clips or b, device=DEV)
    return cfg, eng


def oracle_params(rng, cfg, shape):
    return O.init_params(rng, cfg.num_classes, cfg.frame_encoding_layer, cfg.lstm_hidden, cfg.lstm_layers, shape,
                         classifier=cfg.classifier, well_scaled=True, fusion=cfg.fusion)


@pytest.mark.parametrize("layer,layers,fusion,hid,math", [("fc6", 1, "avg", 8, "f32"), ("fc7", 2, "last", 12, "f32"),
                                                          ("fc8", 1, "avg", 7, "f32"), ("fc6", 2, "state", 9, "f32"),
                                                          ("fc6", 1, "reshape", 8, "f32"), ("fc7", 2, "reshape", 7, "f32"),
                                                          ("fc6", 1, "avg", 8, "bf16x3"), ("fc6", 1, "avg", 8, "bf16x6")])
def test_train_step_small(layer, layers, fusion, hid, math):
    """math="bf16x3": NetConfig.conv_math, the opt-in split-bf16 conv products -- same oracle, same tolerances."""
    rng = np.random.default_rng(5)
    shape, ncls, fpc, b = (67, 67, 3), 7, 3, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, frame_encoding_layer=layer, lstm_hidden=hid, lstm_layers=layers,
                         fusion=fusion, conv_math=math), shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
    lab = rng.integers(0, ncls, b * fpc if fusion == "reshape" else b)     # reshape fusion (tf_util.py:26-27): a logits row per frame
    onehot = O.labels_to_one_hot([[l] for l in lab], ncls)
    x = frames.astype(np.float32) - MEAN
    newp, loss, gn, acc, logits, grads = O.lrcn_train_step(p, x, onehot, fpc, lr=0.01, clip_norm=0.5, final_layer=layer,
                                                          lstm_layers=layers, fusion=fusion)
    fd = torch.tensor(frames, device=DEV)
    got_fwd = eng.forward_u8(fd, MEAN).cpu().numpy()
    assert got_fwd.shape == logits.shape
    np.testing.assert_allclose(got_fwd, logits, rtol=1e-3, atol=1e-3)
    out = eng.train_step_u8(fd, torch.tensor(onehot, device=DEV), lr=0.01, clip_norm=0.5, mean_bgr=MEAN)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    assert abs(out["grad_norm"] - gn) < 1e-3 * gn
    assert out["accuracy"] == O.accuracy(got_fwd, onehot)     # argmax of the device logits (near-ties may differ from fp64)
    g = eng.get_grads()
    for k in p:
        scale = np.abs(grads[k]).max() + 1e-12
        if math in ("f32", "bf16x6"):                # bf16x6 (error at fp32 rounding level) is held to the fp32 bound
            np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=2e-4 * scale, err_msg="grad " + k)
        else:
            # split products move a pre-activation by ~5e-6 relative, enough to flip a ReLU / arg-max decision that the fp32
            # path happens to share with the oracle here (measured: 8 of 34,848 conv1W elements off by 3e-4 of the largest
            # element); bound: 1e-3 of the largest element and 1e-3 relative L2 per tensor
            np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=1e-3 * scale, err_msg="grad " + k)
            assert np.linalg.norm(g[k] - grads[k]) <= 1e-3 * np.linalg.norm(grads[k]) + 1e-12, k
    got = eng.get_params()
    for k in p:
        np.testing.assert_allclose(got[k], newp[k], rtol=1e-4, atol=1e-5, err_msg="param " + k)
    # the reference's own feed format (fp32 NHWC placeholder) gives the same logits
    eng.load_params(p)
    got2 = eng.forward_f32(torch.tensor(x, device=DEV)).cpu().numpy()
    np.testing.assert_allclose(got2, got_fwd, rtol=0, atol=0)


def test_partial_batch_and_crop_mirror():
    """Fewer clips than the engine was sized for (last batch of an epoch, dataset_.py:607-611) and
    the device-side crop/mirror of process_image."""
    rng = np.random.default_rng(9)
    raw, shape, ncls, fpc = (80, 90, 3), (67, 67, 3), 5, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, lstm_hidden=8), shape, 1, max_clips=3)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = rng.integers(0, 256, (fpc,) + raw, dtype=np.uint8)
    cy, cx = np.array([3, 13], np.int32), np.array([20, 0], np.int32)
    mir = np.array([1, 0], np.uint8)
    x = np.stack([O.process_image(frames[i], shape, (cy[i], cx[i]), MEAN, bool(mir[i])) for i in range(fpc)])
    want, _ = O.lrcn_forward(p, x, fpc)
    got = eng.forward_u8(torch.tensor(frames, device=DEV), MEAN, torch.tensor(cy, device=DEV), torch.tensor(cx, device=DEV),
                         torch.tensor(mir, device=DEV)).cpu().numpy()
    assert got.shape == (1, ncls)
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3)
    with pytest.raises(Exception):
        eng.forward_u8(torch.tensor(frames[:1], device=DEV))          # not a multiple of fpc


@pytest.mark.parametrize("ff", [None, ("early", "avg"), ("late", "avg")])
def test_single_frame_fc_classifier(ff):
    """Config 1 of BASELINE.json: frame-level AlexNet (fc8 logits) with classifier fc and frame fusion."""
    rng = np.random.default_rng(2)
    shape, ncls, fpc, b = (67, 67, 3), 6, 3, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, frame_encoding_layer="fc7", classifier="fc", frame_fusion=ff), shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    x = (rng.integers(0, 256, (b * fpc,) + shape).astype(np.float32) - MEAN)
    want, _ = O.lrcn_forward(p, x, fpc, final_layer="fc7", classifier="fc", frame_fusion=ff)
    got = eng.forward_f32(torch.tensor(x, device=DEV)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("layer,ff", [("fc7", ("early", "avg")), ("fc7", ("late", "last")), ("fc8", ("late", "avg")), ("fc6", None)])
def test_fc_classifier_train_step(layer, ff):
    """Config 1 of BASELINE.json as a TRAIN step: frame-level AlexNet, classifier fc, early / late / no frame fusion
    (model.py:103-119,149-151) -- loss, gradients and the clipped-SGD update against the oracle."""
    rng = np.random.default_rng(12)
    shape, ncls, fpc, b = (67, 67, 3), 6, 3, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, frame_encoding_layer=layer, classifier="fc", frame_fusion=ff), shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
    rows = b if ff else b * fpc
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, rows)], ncls)
    x = frames.astype(np.float32) - MEAN
    newp, loss, gn, acc, logits, grads = O.lrcn_train_step(p, x, onehot, fpc, lr=0.01, clip_norm=0.5, final_layer=layer,
                                                          classifier="fc", frame_fusion=ff)
    out = eng.train_step_u8(torch.tensor(frames, device=DEV), torch.tensor(onehot, device=DEV), lr=0.01, clip_norm=0.5, mean_bgr=MEAN)
    np.testing.assert_allclose(eng.logits_host(), logits, rtol=1e-3, atol=1e-3)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss)) and abs(out["grad_norm"] - gn) < 1e-3 * gn
    g = eng.get_grads()
    for k in p:
        scale = np.abs(grads[k]).max() + 1e-12
        np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=2e-4 * scale, err_msg="grad " + k)
    got = eng.get_params()
    for k in p:
        np.testing.assert_allclose(got[k], newp[k], rtol=1e-4, atol=1e-5, err_msg="param " + k)


def test_plain_bf16_conv_mode_runs_close():
    """NetConfig.conv_math = "bf16" (BASELINE config 5: bf16-MFMA conv path): reduced precision by design -- logits within 3e-2
    of the oracle (fp32 path: 1e-3), loss within 1 %, and the step trains."""
    rng = np.random.default_rng(5)
    shape, ncls, fpc, b = (67, 67, 3), 7, 3, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, lstm_hidden=8, conv_math="bf16"), shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, b)], ncls)
    x = frames.astype(np.float32) - MEAN
    _, loss, gn, _, logits, _ = O.lrcn_train_step(p, x, onehot, fpc, lr=0.01, clip_norm=0.5)
    fd = torch.tensor(frames, device=DEV)
    got = eng.forward_u8(fd, MEAN).cpu().numpy()
    np.testing.assert_allclose(got, logits, rtol=3e-2, atol=3e-2)
    assert np.abs(got - logits).max() > 1e-6                      # not the fp32 path
    out = eng.train_step_u8(fd, torch.tensor(onehot, device=DEV), lr=0.01, clip_norm=0.5, mean_bgr=MEAN)
    assert abs(out["loss"] - loss) < 1e-2 * max(1, abs(loss)) and abs(out["grad_norm"] - gn) < 5e-2 * gn


_FULL = {}


def full_geometry_case():
    """Inputs and the oracle's step for the full-geometry test, computed once for both arithmetic modes."""
    if not _FULL:
        from vltf_amd.engine import NetConfig
        rng = np.random.default_rng(1)
        shape, ncls, fpc, b = (227, 227, 3), 101, 4, 2
        p = oracle_params(rng, NetConfig(image_shape=shape, num_classes=ncls, fpc=fpc), shape)
        frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
        onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, b)], ncls)
        x = frames.astype(np.float32) - MEAN
        _FULL["case"] = (p, frames, onehot, O.lrcn_train_step(p, x, onehot, fpc, lr=1e-3, clip_norm=10.0, chunk=4))
    return _FULL["case"]


@pytest.mark.parametrize("math", ["f32", "bf16x3", "bf16x6"])
def test_full_geometry_logits_and_step(math):
    """227x227x3, 2 clips x 4 frames, fc6 -> LSTM(256) -> 101 classes: the real layer shapes; math="bf16x3" is the opt-in
    split-product conv arithmetic (NetConfig.conv_math) held to the same bounds."""
    shape, ncls, fpc, b = (227, 227, 3), 101, 4, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, conv_math=math), shape, b)
    p, frames, onehot, (newp, loss, gn, acc, logits, grads) = full_geometry_case()
    eng.load_params(p)
    out = eng.train_step_u8(torch.tensor(frames, device=DEV), torch.tensor(onehot, device=DEV), lr=1e-3, clip_norm=10.0,
                            mean_bgr=MEAN)
    np.testing.assert_allclose(eng.logits_host(), logits, rtol=1e-3, atol=1e-3)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss)) and abs(out["grad_norm"] - gn) < 1e-3 * gn
    g = eng.get_grads()
    # deep chain in fp32 vs fp64: a near-tie can flip a pool arg-max / ReLU gate, which moves gradient elements
    # discretely, so bound the relative L2 error per tensor instead of every element.  Measured on this very case
    # (tools/grad_flip_noise.py, fp64 oracle with the weights perturbed by 6e-8 relative): no flip -> 3e-6 everywhere;
    # ONE flip -> 8e-3 on conv1W, 6e-3 on conv1b, 4-5e-3 on conv2W/b, 3e-3 on conv3b, nothing above the flipped gate.
    # The oracle's own fp32 run lands at 5.2e-3 / 1.2e-3 / 6e-4.  Which gates flip depends on the (valid) summation
    # order of the kernels, so conv-stack tensors get room for ~3 flips; tensors above pool5 cannot be hit and stay tight.
    for k in p:
        err = np.linalg.norm((g[k] - grads[k]).ravel()) / (np.linalg.norm(grads[k].ravel()) + 1e-30)
        bound = 2.5e-2 if k == "dcnn/conv1W" else (1.8e-2 if k.startswith("dcnn/conv") else 1e-3)
        assert err < bound, "grad %s: relative L2 error %.3e" % (k, err)


def test_adam_steps_match_oracle():
    """defs.optim.adam (train.py:205-206): three clipped Adam steps against the oracle's TF-1.x Adam on the oracle's gradients."""
    rng = np.random.default_rng(21)
    shape, ncls, fpc, b = (67, 67, 3), 5, 2, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, lstm_hidden=8, optimizer="adam"), shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, b)], ncls)
    x = frames.astype(np.float32) - MEAN
    state, lr, clip = {}, 1e-3, 0.5
    fd, od = torch.tensor(frames, device=DEV), torch.tensor(onehot, device=DEV)
    tiny = {k: np.zeros(v.shape, bool) for k, v in p.items()}
    for step in range(3):
        _, loss, gn, _, _, grads = O.lrcn_train_step(p, x, onehot, fpc, lr=0.0, clip_norm=clip)      # gradients at the current p
        clipped, _ = O.clip_by_global_norm(grads, clip)
        p = O.adam_update(p, grads, state, lr, clip)
        out = eng.train_step_u8(fd, od, lr=lr, clip_norm=clip, mean_bgr=MEAN)
        assert abs(out["loss"] - loss) < 2e-4 * max(1, abs(loss)) and abs(out["grad_norm"] - gn) < 2e-3 * gn, step
        got = eng.get_params()
        for k in p:
            # One Adam step moves a weight by ~lr * g / (|g| + eps') whatever the gradient's size (eps' = 1e-8/sqrt(1-beta2) = 3e-7 at
            # step 1), so the parameters are compared at 2 % of lr per step -- except where |g| is within ~30x of eps': there
            # d(update)/dg = lr/eps' and fp32 rounding of the gradient (1e-8 absolute) is already several % of lr (measured
            # 78 of 131k LSTM weights at 3.4 % of lr); those only have to stay within the step size itself.
            tiny[k] |= np.abs(clipped[k]) < 1e-5
            d = np.abs(got[k] - p[k])
            assert d[~tiny[k]].max(initial=0) <= 0.02 * lr * (step + 1) + 1e-6, (step, k, d[~tiny[k]].max())
            assert d.max() <= lr * (step + 1) + 1e-6, (step, k, d.max())


def test_adam_and_dropout_run():
    rng = np.random.default_rng(3)
    shape, ncls, fpc, b = (67, 67, 3), 5, 2, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, lstm_hidden=8, dropout_keep_prob=0.5, optimizer="adam"), shape, b)
    eng.load_params(oracle_params(rng, cfg, shape))
    frames = torch.tensor(rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8), device=DEV)
    onehot = torch.tensor(O.labels_to_one_hot([[0], [1]], ncls), device=DEV)
    losses = [eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN)["loss"] for _ in range(8)]
    assert all(np.isfinite(losses)) and min(losses[4:]) < losses[0]


def test_two_stream_backward_equals_one_stream(monkeypatch):
    """engine._side_stream: weight gradients (and fc6's input gradient) on a second HIP stream beside the input gradients.  Same kernels
    on the same buffers, so three steps must leave exactly the parameters of the one-stream schedule -- a missing wait would not."""
    rng = np.random.default_rng(9)
    shape, ncls, fpc, b = (99, 83, 3), 6, 4, 3
    frames = torch.tensor(rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8), device=DEV)
    onehot = torch.tensor(O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, b)], ncls), device=DEV)
    results = []
    for mode in ("0", ""):                         # one stream; the default = two
        monkeypatch.setenv("VLTF_WGRAD_STREAM", mode) if mode else monkeypatch.delenv("VLTF_WGRAD_STREAM", raising=False)
        cfg, eng = make(dict(num_classes=ncls, fpc=fpc, lstm_hidden=16), shape, b)
        eng.load_params(oracle_params(np.random.default_rng(4), cfg, shape))
        outs = [eng.train_step_u8(frames, onehot, lr=0.01, clip_norm=1.0, mean_bgr=MEAN) for _ in range(3)]
        assert (eng._side_stream() is not None) == (mode != "0")
        results.append((eng.get_params(), [(o["loss"], o["grad_norm"]) for o in outs]))
    (p0, o0), (p1, o1) = results
    assert o0 == o1
    for k in p0:
        assert np.array_equal(p0[k], p1[k]), k


def test_engine_reports_an_lstm_cluster_timeout():
    """The product path reads the cluster LSTM's sticky time-out word wherever it fetches results: a step whose recurrence gave up
    on a peer workgroup raises instead of returning wrong numbers (forced with the library's test hooks), and the engine is
    usable again afterwards."""
    from vltf_amd import ops
    from vltf_amd._ffi import VltfError
    rng = np.random.default_rng(6)
    shape, ncls, fpc, b = (67, 67, 3), 7, 4, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, frame_encoding_layer="fc6", lstm_hidden=32, lstm_layers=1, fusion="avg"), shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = torch.tensor(rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8), device=DEV)
    onehot = torch.tensor(O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, b)], ncls), device=DEV)
    try:
        ops.lstm_seq_test_hooks(spin_limit=64, mute_workgroup=1)
        with pytest.raises(VltfError, match="timed out"):
            eng.train_step_u8(frames, onehot, lr=0.0, clip_norm=0.0, mean_bgr=MEAN)
        eng.forward_u8(frames, MEAN)
        with pytest.raises(VltfError, match="timed out"):
            eng.logits_host()
    finally:
        ops.lstm_seq_test_hooks()
    eng.load_params(p)
    out = eng.train_step_u8(frames, onehot, lr=0.0, clip_norm=0.0, mean_bgr=MEAN)
    assert np.isfinite(out["loss"])


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_a_timed_out_step_does_not_reach_the_weights(optimizer):
    """fetch=False steps never read the status on the host; a step whose LSTM cluster launch timed out must still not be applied: the
    optimizer launch drops it on the device (ops.step_guard -> the `skip` word of vl_sgd_apply / vl_adam_apply), the next fetch raises
    (round 3 applied such a step and went on training on its results: the advisor's finding)."""
    from vltf_amd import ops
    from vltf_amd._ffi import VltfError
    rng = np.random.default_rng(8)
    shape, ncls, fpc, b = (67, 67, 3), 7, 4, 2
    cfg, eng = make(dict(num_classes=ncls, fpc=fpc, frame_encoding_layer="fc6", lstm_hidden=32, lstm_layers=1, fusion="avg", optimizer=optimizer),
                    shape, b)
    p = oracle_params(rng, cfg, shape)
    eng.load_params(p)
    frames = torch.tensor(rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8), device=DEV)
    onehot = torch.tensor(O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, b)], ncls), device=DEV)
    before = eng.w.clone()
    try:
        ops.lstm_seq_test_hooks(spin_limit=64, mute_workgroup=1)
        assert eng.train_step_u8(frames, onehot, lr=0.1, clip_norm=0.0, mean_bgr=MEAN, fetch=False) is None
        torch.cuda.synchronize()
        assert torch.equal(eng.w, before), "a step whose recurrence timed out changed the weights"
        with pytest.raises(VltfError, match="timed out"):
            eng.check_status()
    finally:
        ops.lstm_seq_test_hooks()
    eng.train_step_u8(frames, onehot, lr=0.1, clip_norm=0.0, mean_bgr=MEAN, fetch=False)      # a healthy step IS applied
    torch.cuda.synchronize()
    assert not torch.equal(eng.w, before)
    eng.check_status()


def test_two_stream_backward_of_the_lstm_pipeline_under_allocator_churn():
    """The single-pipeline LRCN (dcnn -> LSTM, 1 and 2 layers) on its default two-stream backward: head, LSTM, fc6 and conv parameter
    gradients and the dgrad weight transposes run on the second stream.  Many FRESH engines with the caching allocator perturbed
    between them (blocks of large stale values), every gradient of every step against the oracle -- the kind of test that found a
    one-in-a-hundred missing dependency in round 3 (tests/test_graph_gpu.py::test_two_stream_backward_under_allocator_churn)."""
    rng = np.random.default_rng(11)
    shape, ncls, fpc, b = (67, 67, 3), 7, 3, 2
    frames = rng.integers(0, 256, (b * fpc,) + shape, dtype=np.uint8)
    lab = rng.integers(0, ncls, b)
    onehot = O.labels_to_one_hot([[l] for l in lab], ncls)
    x = frames.astype(np.float32) - MEAN
    fd, od = torch.tensor(frames, device=DEV), torch.tensor(onehot, device=DEV)
    want, junk = {}, []
    for it in range(60):
        layers = 1 + it % 2
        junk.append(torch.full((int(rng.integers(1, 64)) << 18,), 1e3, device=DEV))
        if len(junk) > 6:
            del junk[int(rng.integers(0, len(junk)))]
        cfg, eng = make(dict(num_classes=ncls, fpc=fpc, lstm_hidden=8, lstm_layers=layers), shape, b)
        if layers not in want:
            p = oracle_params(np.random.default_rng(5 + layers), cfg, shape)
            want[layers] = (p,) + O.lrcn_train_step(p, x, onehot, fpc, lr=0.01, clip_norm=0.5, lstm_layers=layers)
        p, newp, loss, gn, acc, logits, grads = want[layers]
        eng.load_params(p)
        out = eng.train_step_u8(fd, od, lr=0.01, clip_norm=0.5, mean_bgr=MEAN)
        assert (eng._side_stream() is not None) == (os.environ.get("VLTF_WGRAD_STREAM", "") != "0")
        assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss)), it
        g = eng.get_grads()
        for k in p:
            scale = np.abs(grads[k]).max() + 1e-12
            np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=2e-4 * scale, err_msg="iteration %d, grad %s" % (it, k))
        del eng
