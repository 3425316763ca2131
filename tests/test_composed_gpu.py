"""BASELINE config 4 -- the video-description encoder-decoder assembled from two pipelines (models/model.py:18-162): frames ->
AlexNet -> LSTM (fusion state) => state vector of a decoder LSTM over word vectors, per-step logits (fusion reshape), word-level
softmax cross-entropy -- on the device against the CPU oracle (oracle.lrcn_oracle.encdec_* / lstm_classifier_* /
tensor_list_fusion), forward logits and every gradient including those that reach the encoder through the initial-state path.
Tolerances as for the single-pipeline engine tests: logits 1e-3, gradients 2e-3 relative / 2e-4 of the largest element."""
import numpy as np
import pytest
import torch

from oracle import lrcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def build(shape, V, enc_h, enc_layers, dec_h, dec_layers, fusion, E, Tf, Tw, b, seed, input_fusion=None, representation="nop",
          fc_out=None, layer="fc6", ratio=1):
    from vltf_amd.composed import ComposedEngine, HeadConfig
    from vltf_amd.engine import NetConfig
    rng = np.random.default_rng(seed)
    enc_cfg = NetConfig(image_shape=shape, num_classes=V, fpc=Tf, frame_encoding_layer=layer, classifier="lstm", lstm_hidden=enc_h,
                        lstm_layers=enc_layers, fusion="state")
    head = HeadConfig(in_dim=E, fpc=Tw, num_classes=V, lstm_hidden=dec_h, lstm_layers=dec_layers, fusion=fusion,
                      representation=representation, fc_output_dim=fc_out, input_fusion=input_fusion, cpv_ratio=ratio)
    eng = ComposedEngine(enc_cfg, head, max_clips=b, device=DEV)
    pe = O.init_params(rng, V, layer, enc_h, enc_layers, shape, well_scaled=True, fusion="state")
    p = {"enc/" + k: v for k, v in pe.items()}
    fused = E + V if input_fusion == "concat" else E            # the input fusion comes first (model.py:69-73) ...
    seq_dim = fc_out if representation == "fc" else fused         # ... then the representation (model.py:81-96)
    p.update(O.init_lstm_classifier_params(rng, "dec/", seq_dim, dec_h, dec_layers, fusion, V,
                                           state_dim=V if input_fusion is None else None, well_scaled=True))
    if representation == "fc":
        p["dec/fc_convert_w"] = O.truncated_normal(rng, (fused, fc_out), np.sqrt(2.0 / fused))
        p["dec/fc_convert_b"] = np.full(fc_out, 0.1, np.float32)
    eng.load_params(p)
    frames = rng.integers(0, 256, (b * Tf,) + shape, dtype=np.uint8)
    words = rng.standard_normal((b * ratio * Tw, E)).astype(np.float32)
    return eng, p, frames, words, rng


def check_grads(g, want, p):
    for k in p:
        scale = np.abs(want[k]).max() + 1e-12
        np.testing.assert_allclose(g[k], want[k], rtol=2e-3, atol=2e-4 * scale, err_msg="grad " + k)


@pytest.mark.parametrize("fusion,enc_layers,dec_layers,enc_h,dec_h", [("reshape", 1, 1, 8, 10), ("reshape", 2, 2, 9, 9),
                                                                      ("avg", 1, 2, 8, 11), ("state", 1, 1, 11, 6)])
def test_encoder_decoder_train_step(fusion, enc_layers, dec_layers, enc_h, dec_h):
    shape, V, E, Tf, Tw, b = (67, 67, 3), 11, 6, 3, 5, 2
    eng, p, frames, words, rng = build(shape, V, enc_h, enc_layers, dec_h, dec_layers, fusion, E, Tf, Tw, b, seed=31)
    rows = b * Tw if fusion == "reshape" else b
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, V, rows)], V)
    x = frames.astype(np.float32) - MEAN
    enc, dec = dict(layer="fc6", layers=enc_layers), dict(layers=dec_layers, fusion=fusion)
    logits, cache = O.encdec_forward(p, x, words, Tf, Tw, enc, dec, V)
    loss, dlogits = O.softmax_xent_mean(logits, onehot)
    want = O.encdec_backward(p, cache, dlogits, Tf, enc)
    fd, wd, od = torch.tensor(frames, device=DEV), torch.tensor(words, device=DEV), torch.tensor(onehot, device=DEV)
    got = eng.forward(fd, wd, MEAN).cpu().numpy()
    assert got.shape == (rows, V)
    np.testing.assert_allclose(got, logits, rtol=1e-3, atol=1e-3)
    out = eng.train_step(fd, wd, od, lr=0.01, clip_norm=0.5, mean_bgr=MEAN)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    clipped, gn = O.clip_by_global_norm(want, 0.5)
    assert abs(out["grad_norm"] - gn) < 1e-3 * gn
    check_grads(eng.get_grads(), want, p)
    newp = eng.get_params()
    for k in p:
        np.testing.assert_allclose(newp[k], p[k].astype(np.float64) - 0.01 * clipped[k], rtol=1e-4, atol=1e-5, err_msg="param " + k)
    assert not any(("enc/" in k) and np.abs(want[k]).max() == 0 for k in p if k.endswith("W")), "no gradient reached the encoder"


def test_config4_full_geometry():
    """BASELINE config 4 at its real layer shapes: 2 clips x 4 frames of 227x227x3 through AlexNet(fc6) + LSTM(256, state), the state
    vector of a 256-unit decoder over 21 word vectors of 300 dimensions, per-step logits over 1000 classes; one train step against the
    oracle (logits 1e-3; gradients by relative L2 per tensor: 1e-3 above pool5, where no ReLU / arg-max flip can reach, and the
    un-gated conv-stack bound of tests/test_engine_gpu.py below it)."""
    shape, V, E, Tf, Tw, b, H = (227, 227, 3), 1000, 300, 4, 21, 2, 256
    eng, p, frames, words, rng = build(shape, V, H, 1, H, 1, "reshape", E, Tf, Tw, b, seed=41)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, V, b * Tw)], V)
    x = frames.astype(np.float32) - MEAN
    enc, dec = dict(layer="fc6", layers=1), dict(layers=1, fusion="reshape")
    logits, cache = O.encdec_forward(p, x, words, Tf, Tw, enc, dec, V, chunk=4)
    loss, dlogits = O.softmax_xent_mean(logits, onehot)
    want = O.encdec_backward(p, cache, dlogits, Tf, enc)
    fd, wd, od = torch.tensor(frames, device=DEV), torch.tensor(words, device=DEV), torch.tensor(onehot, device=DEV)
    out = eng.train_step(fd, wd, od, lr=0.0, clip_norm=0.0, mean_bgr=MEAN)
    np.testing.assert_allclose(eng.logits_host(), logits, rtol=1e-3, atol=1e-3)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    g = eng.get_grads()
    for k in p:
        err = np.linalg.norm((g[k] - want[k]).ravel()) / (np.linalg.norm(want[k].ravel()) + 1e-30)
        bound = 2.5e-2 if k.startswith("enc/dcnn/conv") else 1e-3
        assert err < bound, "grad %s: relative L2 error %.3e" % (k, err)


@pytest.mark.parametrize("input_fusion,representation,ratio", [("concat", "nop", 2), ("ibias", "fc", 1), (None, "fc", 2), ("ibias", "fc", 3)])
def test_input_fusion_replication_and_fc_representation(input_fusion, representation, ratio):
    """apply_tensor_list_fusion concat (the vec_seq_concat branch: clips-per-video ratio > 1) / ibias of a word sequence with the
    per-video vector of pipeline 1, replicate_auxilliary_tensor at ratio > 1 (also on the state path, model.py:131-134), and
    representation fc (convert_dim_fc) -- applied AFTER the input fusion, as build_pipeline orders them (model.py:69-96): under
    ibias the word vectors must therefore be as wide as pipeline 1's output, and the fc maps the T + 1 fused steps."""
    shape, V, Tf, Tw, b1, H = (67, 67, 3), 7, 2, 4, 2, 8
    E = V if input_fusion == "ibias" else 5                   # ibias: the vector becomes one more time step of the same width
    fc_out = 6 if representation == "fc" else None
    eng, p, frames, words, rng = build(shape, V, 6, 1, H, 1, "reshape", E, Tf, Tw, b1, seed=5, input_fusion=input_fusion,
                                       representation=representation, fc_out=fc_out, ratio=ratio)
    pipes = [("enc", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier="lstm", lstm_params=[6, 1, "state"])),
             ("dec", dict(input=["aux", "enc"], input_fusion=input_fusion, representation=representation, fc_output_dim=fc_out,
                          classifier="lstm", lstm_params=[H, 1, "reshape"]))]
    ds = {"main": dict(cpv=1, fpc=Tf), "aux": dict(cpv=ratio, fpc=Tw)}
    x = frames.astype(np.float32) - MEAN
    logits, cache = O.model_forward(p, pipes, ds, {"main": x, "aux": words}, V)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, V, logits.shape[0])], V)
    loss, dlogits = O.softmax_xent_mean(logits, onehot)
    g = O.model_backward(p, cache, dlogits)
    fd, wd, od = torch.tensor(frames, device=DEV), torch.tensor(words, device=DEV), torch.tensor(onehot, device=DEV)
    np.testing.assert_allclose(eng.forward(fd, wd, MEAN).cpu().numpy(), logits, rtol=1e-3, atol=1e-3)
    out = eng.train_step(fd, wd, od, lr=0.0, clip_norm=0.0, mean_bgr=MEAN)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    check_grads(eng.get_grads(), g, p)


def test_concat_at_ratio_one_is_refused():
    from vltf_amd._ffi import VltfError
    with pytest.raises(VltfError, match="ratio"):
        build((67, 67, 3), 7, 6, 1, 8, 1, "reshape", 5, 2, 4, 2, seed=1, input_fusion="concat")


def test_tensor_list_ops(ops=None):
    """vl_copy2d / vl_eltwise2 / vl_max2_grad against apply_tensor_list_fusion avg | maximum | concat (equal rows) and
    replicate_auxilliary_tensor (tf_util.py:136-192)."""
    from vltf_amd import ops
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((6, 5)).astype(np.float32), rng.standard_normal((6, 5)).astype(np.float32)
    ad, bd = torch.tensor(a, device=DEV), torch.tensor(b, device=DEV)
    out = torch.empty_like(ad)
    for method, op in (("avg", "avg"), ("maximum", "maximum")):
        want, _, _, _, cache = O.tensor_list_fusion([a, b], method, [5, 5], [1, 1], [1, 1])
        ops.eltwise2(ad, bd, out, op)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6)
    d = rng.standard_normal((6, 5)).astype(np.float32)
    da, db = torch.empty_like(ad), torch.empty_like(ad)
    ops.max2_grad(ad, bd, torch.tensor(d, device=DEV), da, db)
    wa, wb = O.tensor_list_fusion_grad(cache, d)
    np.testing.assert_array_equal(da.cpu().numpy(), wa)
    np.testing.assert_array_equal(db.cpu().numpy(), wb)
    # concat with equal rows = two column-block copies
    c = rng.standard_normal((6, 3)).astype(np.float32)
    want, dim, _, _, _ = O.tensor_list_fusion([a, c], "concat", [5, 3], [1, 1], [1, 1])
    cat = torch.empty((6, 8), device=DEV)
    ops.copy2d(ad, cat, 6, 5, src_ld=5, dst_ld=8)
    ops.copy2d(torch.tensor(c, device=DEV), cat[:, 5:], 6, 3, src_ld=3, dst_ld=8)
    np.testing.assert_array_equal(cat.cpu().numpy(), want)
    # replicate_auxilliary_tensor: the whole batch repeated tile_num times (src_ld = 0 repeats the one "row")
    rep = torch.empty((3 * 6, 5), device=DEV)
    ops.copy2d(ad, rep, 3, 30, src_ld=0, dst_ld=30)
    np.testing.assert_array_equal(rep.cpu().numpy(), O.replicate_auxilliary_tensor(a, 3))


def _dp_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vltf_amd import dp
    from vltf_amd.composed import ComposedEngine, HeadConfig
    from vltf_amd.engine import NetConfig
    dp.init_from_env(backend="gloo")
    shape, V, E, Tf, Tw, clips = (67, 67, 3), 7, 5, 2, 3, 4
    rng = np.random.default_rng(17)                                      # identical on every rank
    enc_cfg = NetConfig(image_shape=shape, num_classes=V, fpc=Tf, classifier="lstm", lstm_hidden=6, fusion="state")
    head = HeadConfig(in_dim=E, fpc=Tw, num_classes=V, lstm_hidden=8, lstm_layers=1, fusion="reshape")
    pe = O.init_params(rng, V, "fc6", 6, 1, shape, well_scaled=True, fusion="state")
    p = {"enc/" + k: v for k, v in pe.items()}
    p.update(O.init_lstm_classifier_params(rng, "dec/", E, 8, 1, "reshape", V, state_dim=V, well_scaled=True))
    frames = rng.integers(0, 256, (clips * Tf,) + shape, dtype=np.uint8)
    words = rng.standard_normal((clips * Tw, E)).astype(np.float32)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, V, clips * Tw)], V)
    lo, hi = dp.shard_range(clips, rank, world)
    gar = dp.GradAllReduce()
    eng = ComposedEngine(enc_cfg, head, max_clips=hi - lo, device=DEV, dp=gar)
    eng.load_params(p)
    gar.broadcast_params(eng.w)
    out = eng.train_step(torch.tensor(frames[lo * Tf:hi * Tf], device=DEV), torch.tensor(words[lo * Tw:hi * Tw], device=DEV),
                         torch.tensor(onehot[lo * Tw:hi * Tw], device=DEV), lr=0.05, clip_norm=0.5, mean_bgr=MEAN,
                         global_rows=clips * Tw)
    got = eng.get_params()
    if rank == 0:
        ref = ComposedEngine(enc_cfg, head, max_clips=clips, device=DEV)
        ref.load_params(p)
        want_out = ref.train_step(torch.tensor(frames, device=DEV), torch.tensor(words, device=DEV), torch.tensor(onehot, device=DEV),
                                  lr=0.05, clip_norm=0.5, mean_bgr=MEAN)
        want = ref.get_params()
        err = {k: float(np.abs(got[k] - want[k]).max() / (np.abs(want[k] - p[k]).max() + 1e-12)) for k in want}
        q.put(("ok", max(err.values()), abs(out["grad_norm"] - want_out["grad_norm"]) / want_out["grad_norm"], gar.issued, len(eng.grad_chunks)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_composed_step_equals_one_rank():
    """Data parallel over clips for the two-pipeline model: the shared flat gradient buffer goes out in chunks -- pipeline 2's
    first, then pipeline 1's through the offset view of its own chunk list -- and two ranks (gloo, one GPU) end with the
    parameters of one rank stepping on all the clips."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, "rank exited with %s" % pr.exitcode
    tag, worst, gn_err, issued, nchunks = q.get(timeout=10)
    assert tag == "ok" and issued == nchunks and nchunks >= 4, (issued, nchunks)
    assert worst < 1e-3 and gn_err < 1e-5, (worst, gn_err)
