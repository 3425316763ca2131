"""HOST LOGIC of vltf_amd.graph.GraphEngine -- the general pipeline graph of models/model.py:18-162 -- in the build container:
the engine's own Python runs unchanged, the C-ABI kernels and the AlexNet tower are replaced by the torch-CPU stand-ins of
tests/cpu_double.py (test infrastructure), and every logit and gradient must equal the CPU oracle's
(oracle.lrcn_oracle.model_forward / model_backward).  What this pins: stage order, buffer routing, gradient accumulation over
fan-out, variable naming / the flat buffer, pruning of unused pipelines, the refusals.  The kernels are the GPU tests' business."""
import numpy as np
import pytest
import torch

from tests import graph_cases as GC
from tests.cpu_double import install

torch.set_num_threads(4)


def run_case(monkeypatch, name, dp=None):
    Engine = install(monkeypatch)
    case = GC.CASES[name]()
    pipes, ds = GC.specs_and_datasets(case)
    eng = Engine(pipes, ds, case["V"], device="cpu", dp=dp)
    p = eng.init_params(seed=case["seed"], well_scaled=True)
    eng.load_params(p)
    raw, feeds = GC.inputs(case)
    logits, onehot, loss, grads, _ = GC.expect(case, p, feeds)
    dev_feeds = {t: (dict(frames_u8=torch.from_numpy(v), mean_bgr=GC.MEAN) if v.dtype == np.uint8 else torch.from_numpy(v)) for t, v in raw.items()}
    got = eng.forward(dev_feeds).numpy().copy()
    assert got.shape == logits.shape
    np.testing.assert_allclose(got, logits, rtol=1e-4, atol=1e-4)
    out = eng.train_step(dev_feeds, torch.from_numpy(onehot), lr=0.01, clip_norm=0.5)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    clipped, gn = O_clip(grads)
    assert abs(out["grad_norm"] - gn) < 1e-3 * gn
    g = eng.get_grads()
    assert set(g) == set(p)
    for k in p:
        scale = np.abs(grads[k]).max() + 1e-12
        np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=2e-4 * scale, err_msg="grad " + k)
    newp = eng.get_params()
    for k in p:
        np.testing.assert_allclose(newp[k], p[k].astype(np.float64) - 0.01 * clipped[k], rtol=1e-4, atol=1e-5, err_msg="param " + k)
    return eng, case, p, grads


def O_clip(grads):
    from oracle import lrcn_oracle as O
    return O.clip_by_global_norm(grads, 0.5)


@pytest.mark.parametrize("name", sorted(GC.CASES))
def test_graph_matches_oracle_on_the_cpu_double(monkeypatch, name):
    eng, case, p, grads = run_case(monkeypatch, name)
    if name.startswith("two_stream"):
        assert np.abs(grads["rgb/dcnn/conv1W"]).max() > 0 and np.abs(grads["flow/dcnn/conv1W"]).max() > 0
    if name == "fanout":
        assert eng.skipped == ["unused"] and not any(k.startswith("unused/") for k in p)
        assert np.abs(grads["feat/dcnn/fc7W"]).max() > 0
    if name.startswith("fused_frames"):
        assert "dcnn/conv1W" in p               # a single pipeline keeps the unscoped TF names


class _CountingDp:
    """Records the order of the gradient chunks a step hands to the exchange (no communication: world of one)."""
    world = 1

    def __init__(self):
        self.calls = []

    def reduce_async(self, flat, off, cnt):
        self.calls.append((off, cnt))

    def wait(self):
        pass


def test_exchange_chunks_cover_the_buffer_and_wait_for_the_last_lstm(monkeypatch):
    """Data-parallel plumbing: every element of the flat gradient goes out exactly once, in the order backward completes it, and
    nothing is issued before the LAST LSTM backward of the step has been launched (the cluster-form recurrence must not spin under
    an RCCL kernel: the decoder's chunk is held until the encoder's LSTM has run)."""
    dp = _CountingDp()
    eng, case, p, _ = run_case(monkeypatch, "encdec_state", dp=dp)
    calls = dp.calls
    assert sorted(calls) == sorted(eng.grad_chunks)
    covered = sorted(calls)
    assert covered[0][0] == 0 and all(covered[i][0] + covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    assert covered[-1][0] + covered[-1][1] == eng.w.numel()
    assert calls[0] == eng.grad_chunks[0]                    # the decoder's head first -- but only once the encoder's LSTM is through:
    order = []
    orig = eng.by_name["enc"]._lstm_backward
    eng.by_name["enc"]._lstm_backward = lambda d: (order.append("enc_lstm"), orig(d))[1]
    dp.reduce_async = lambda flat, off, cnt: order.append("reduce")
    raw, _ = GC.inputs(case)
    feeds = {t: (dict(frames_u8=torch.from_numpy(v), mean_bgr=GC.MEAN) if v.dtype == np.uint8 else torch.from_numpy(v)) for t, v in raw.items()}
    rows = eng.forward(feeds).shape[0]
    eng.train_step(feeds, torch.zeros((rows, case["V"]), dtype=torch.int32), lr=0.0)
    assert order[0] == "enc_lstm" and order.count("reduce") == len(eng.grad_chunks)


class _SnapshotDp:
    """Keeps what every chunk held at the moment it was handed to the exchange (the CPU double runs synchronously: that IS what an
    all-reduce issued there would read)."""
    world = 1

    def __init__(self):
        self.snaps = []

    def reduce_async(self, flat, off, cnt):
        self.snaps.append((off, cnt, flat[off:off + cnt].clone()))

    def wait(self):
        pass


@pytest.mark.parametrize("name", ["encdec_ibias_r3", "dcnn_with_state", "fanout", "encdec_concat_r2", "two_stream_avg"])
def test_every_gradient_of_a_chunk_is_written_before_the_chunk_is_exchanged(monkeypatch, name):
    """A chunk handed to the exchange must already hold its FINAL gradients.  Round 3's GraphEngine released a pipeline's head chunk
    before the `representation: fc` gradients (fc_convert_w / _b, part of that chunk) had been computed: under data parallelism the
    all-reduce read stale values (the advisor's finding; `encdec_ibias_r3` and `fanout` have such a pipeline).  The gradient buffer
    starts as NaN, so a value exchanged before it was written cannot equal the final one."""
    Engine = install(monkeypatch)
    case = GC.CASES[name]()
    pipes, ds = GC.specs_and_datasets(case)
    dp = _SnapshotDp()
    eng = Engine(pipes, ds, case["V"], device="cpu", dp=dp)
    eng.load_params(eng.init_params(seed=case["seed"], well_scaled=True))
    raw, feeds = GC.inputs(case)
    dev_feeds = {t: (dict(frames_u8=torch.from_numpy(v), mean_bgr=GC.MEAN) if v.dtype == np.uint8 else torch.from_numpy(v)) for t, v in raw.items()}
    rows = eng.forward(dev_feeds).shape[0]
    onehot = np.zeros((rows, case["V"]), np.int32)
    onehot[np.arange(rows), np.random.default_rng(3).integers(0, case["V"], rows)] = 1
    eng.g.fill_(float("nan"))
    eng.train_step(dev_feeds, torch.from_numpy(onehot), lr=0.0, clip_norm=0.0)
    assert sorted((o, c) for o, c, _ in dp.snaps) == sorted(eng.grad_chunks)
    final = eng.g.clone()
    assert not torch.isnan(final).any(), "a gradient was never written"
    for off, cnt, snap in dp.snaps:
        assert torch.equal(snap, final[off:off + cnt]), "chunk (%d, %d) was exchanged before all of its gradients were written" % (off, cnt)


@pytest.mark.parametrize("pipes,msg", [
    ([("a", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier=None, frame_fusion=("late", "avg")))], "late fusion with no classifier"),
    ([("a", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier=None))], "classifier is missing"),
    ([("a", dict(input=["aux"], representation="dcnn", frame_encoding_layer="fc6", classifier="fc"))], "rank"),
    ([("a", dict(input=["nope"], representation="nop", classifier="fc"))], "Could not find a dataset"),
    ([("a", dict(input=["aux"], representation="fc", fc_output_dim=9, classifier="fc"))], "already exists"),
    ([("a", dict(input=["aux"], representation="nop", classifier="lstm", lstm_params=(4, 1, "avg"), frame_fusion=("early", "avg")))], "only with"),
    ([("a", dict(input=["aux", "main"], representation="nop", classifier="fc"))], "neither an input_fusion"),
    ([("a", dict(input=["main"], representation="nop", classifier="fc"))], "cannot take the frame dataset"),
    ([("a", dict(input=["aux", "main"], representation="nop", classifier="lstm", lstm_params=(4, 1, "avg")))], "holds frames"),
    ([("f", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier=None)),
      ("a", dict(input=["aux", "f"], input_fusion="concat", representation="nop", classifier="fc"))], "ratio"),
    ([("f", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier="fc", frame_fusion=("early", "avg"))),
      ("a", dict(input=["aux", "f"], input_fusion="ibias", representation="nop", classifier="lstm", lstm_params=(4, 1, "avg")))], "equal widths"),
])
def test_refusals_follow_the_reference(monkeypatch, pipes, msg):
    from vltf_amd._ffi import VltfError
    Engine = install(monkeypatch)
    case = dict(pipes=pipes, data={"main": dict(mode="video", fpc=3, cpv=1), "aux": dict(mode="vectors", fpc=4, cpv=1, dim=6)}, V=7, items=2)
    specs, ds = GC.specs_and_datasets(case)
    with pytest.raises(VltfError, match=msg):
        Engine(specs, ds, 7, device="cpu")


@pytest.mark.parametrize("name", sorted(GC.FULL_CASES))
def test_full_geometry_fixture_covers_every_variable(name):
    """tests/golden/graph_full.npz (the fp64 oracle's answers for the full-geometry graphs of tests/test_graph_full_gpu.py) holds a
    gradient norm, head and sample for exactly the variables the engine plans for that model -- no device needed for the plan."""
    import os
    from vltf_amd.graph import model_specs
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_full.npz"))
    case = GC.FULL_CASES[name]()
    pipes, ds = GC.specs_and_datasets(case)
    specs = model_specs(pipes, ds, case["V"])
    want = {n for n, _ in specs}
    have = {k.split("/gradnorm/", 1)[1] for k in gold.files if k.startswith(name + "/gradnorm/")}
    assert want == have
    rows = case["items"] * (case["data"]["aux"]["fpc"] if name == "c4_ws" else 1)
    assert gold[name + "/logits"].shape == (rows, case["V"])
    loss, gn, acc = gold[name + "/loss_gn_acc"]
    assert np.isfinite([loss, gn, acc]).all() and gn > 0
    for n, shp in specs:
        assert gold["%s/gradsample/%s" % (name, n)].shape == (64,) and np.isfinite(gold["%s/gradnorm/%s" % (name, n)]).all()
