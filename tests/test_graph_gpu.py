"""The general pipeline graph (models/model.py:18-162; vltf_amd.graph.GraphEngine) ON THE DEVICE against the CPU oracle
(oracle.lrcn_oracle.model_forward / model_backward, pinned by torch autograd in tests/test_oracle.py): the two-stream LRCN
(two dcnn feature pipelines fused by input_fusion avg | maximum | concat into an LSTM), feature pipelines with fan-out, a dcnn
pipeline whose LSTM takes another pipeline's output as its state, one pipeline over fused frame datasets, and the encoder-decoder
forms.  Logits 1e-3, every gradient (into every tower) at the single-pipeline tolerances, the applied update.
tests/test_graph_cpu.py runs the same cases through the engine's host logic on a torch-CPU stand-in of the kernels."""
import numpy as np
import pytest
import torch

from oracle import lrcn_oracle as O
from tests import graph_cases as GC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def device_feeds(raw):
    return {t: (dict(frames_u8=torch.from_numpy(v).to(DEV), mean_bgr=GC.MEAN) if v.dtype == np.uint8 else torch.from_numpy(v).to(DEV))
            for t, v in raw.items()}


@pytest.mark.parametrize("name", sorted(GC.CASES))
def test_graph_train_step_matches_oracle(name):
    from vltf_amd.graph import GraphEngine
    case = GC.CASES[name]()
    pipes, ds = GC.specs_and_datasets(case)
    eng = GraphEngine(pipes, ds, case["V"], device=DEV)
    p = eng.init_params(seed=case["seed"], well_scaled=True)
    eng.load_params(p)
    raw, feeds = GC.inputs(case)
    logits, onehot, loss, grads, _ = GC.expect(case, p, feeds)
    fd = device_feeds(raw)
    got = eng.forward(fd).cpu().numpy()
    assert got.shape == logits.shape
    np.testing.assert_allclose(got, logits, rtol=1e-3, atol=1e-3)
    out = eng.train_step(fd, torch.from_numpy(onehot).to(DEV), lr=0.01, clip_norm=0.5)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    clipped, gn = O.clip_by_global_norm(grads, 0.5)
    assert abs(out["grad_norm"] - gn) < 1e-3 * gn
    g = eng.get_grads()
    assert set(g) == set(p)
    for k in p:
        scale = np.abs(grads[k]).max() + 1e-12
        np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=2e-4 * scale, err_msg="grad " + k)
    newp = eng.get_params()
    for k in p:
        np.testing.assert_allclose(newp[k], p[k].astype(np.float64) - 0.01 * clipped[k], rtol=1e-4, atol=1e-5, err_msg="param " + k)
    towers = [k for k in p if k.endswith("dcnn/conv1W")]
    assert towers and all(np.abs(grads[k]).max() > 0 for k in towers), "no gradient reached a tower"


def test_two_stream_full_geometry():
    """The two-stream LRCN at AlexNet's real layer shapes: 2 clips x 4 frames of 227x227x3 per stream, fc6 features averaged into
    LSTM(256) -> 101 classes; logits 1e-3, gradients by relative L2 per tensor (1e-3 above the towers' pool5, the un-gated
    conv-stack bound of tests/test_engine_gpu.py below it: ReLU / arg-max near-ties may fall the other way in fp32)."""
    from vltf_amd.graph import DatasetInfo, GraphEngine, PipelineSpec
    shape, V, T, b, H = (227, 227, 3), 101, 4, 2, 256
    pipes = [("rgb", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier=None)),
             ("flow", dict(input=["aux"], representation="dcnn", frame_encoding_layer="fc6", classifier=None)),
             ("fuse", dict(input=["rgb", "flow"], input_fusion="avg", representation="nop", classifier="lstm", lstm_params=[H, 1, "avg"]))]
    ds = {t: DatasetInfo("video", T, 1, b, image_shape=shape) for t in ("main", "aux")}
    eng = GraphEngine([PipelineSpec(name=n, **{k: (tuple(v) if isinstance(v, list) and k != "input" else v) for k, v in s.items()}) for n, s in pipes],
                      ds, V, device=DEV)
    p = eng.init_params(seed=3, well_scaled=True)
    eng.load_params(p)
    rng = np.random.default_rng(31)
    raw = {t: rng.integers(0, 256, (b * T,) + shape, dtype=np.uint8) for t in ("main", "aux")}
    feeds = {t: v.astype(np.float32) - GC.MEAN for t, v in raw.items()}
    logits, cache = O.model_forward(p, pipes, {t: dict(cpv=1, fpc=T) for t in ds}, feeds, V, chunk=4)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, V, b)], V)
    loss, dlogits = O.softmax_xent_mean(logits, onehot)
    want = O.model_backward(p, cache, dlogits)
    out = eng.train_step(device_feeds(raw), torch.from_numpy(onehot).to(DEV), lr=0.0, clip_norm=0.0)
    np.testing.assert_allclose(eng.logits_host(), logits, rtol=1e-3, atol=1e-3)
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss))
    g = eng.get_grads()
    for k in p:
        err = np.linalg.norm((g[k] - want[k]).ravel()) / (np.linalg.norm(want[k].ravel()) + 1e-30)
        assert err < (2.5e-2 if "/dcnn/conv" in k else 1e-3), "grad %s: relative L2 error %.3e" % (k, err)


def test_two_stream_on_the_bf16_conv_path():
    """The graph's towers on the packed-bf16 conv path (conv_math="bf16", BASELINE config 5's arithmetic; heads' large GEMMs follow the
    mode): reduced precision by design -- logits within 5e-2 of the fp64 oracle, loss within 1e-2, and a finite step that moves both towers."""
    from vltf_amd.graph import GraphEngine
    case = GC.CASES["two_stream_avg"]()
    pipes, ds = GC.specs_and_datasets(case)
    eng = GraphEngine(pipes, ds, case["V"], device=DEV, conv_math="bf16")
    p = eng.init_params(seed=case["seed"], well_scaled=True)
    eng.load_params(p)
    raw, feeds = GC.inputs(case)
    logits, onehot, loss, grads, _ = GC.expect(case, p, feeds)
    out = eng.train_step(device_feeds(raw), torch.from_numpy(onehot).to(DEV), lr=0.01, clip_norm=0.5)
    assert np.abs(eng.logits_host() - logits).max() < 5e-2
    assert abs(out["loss"] - loss) < 1e-2 * max(1, abs(loss)) and np.isfinite(out["grad_norm"])
    g = eng.get_grads()
    for k in ("rgb/dcnn/conv1W", "flow/dcnn/conv1W", "rgb/dcnn/fc6W", "fuse/output_fc_w"):
        err = np.linalg.norm((g[k] - grads[k]).ravel()) / np.linalg.norm(grads[k].ravel())
        assert err < 0.2, "grad %s: relative L2 %.3e" % (k, err)


def test_tie_semantics_of_the_maximum_fusion():
    """vl_fuse_n / vl_fuse_n_grad / vl_max2_grad: inputs equal to the maximum share the gradient evenly (tf.reduce_max's
    _MinOrMaxGrad) -- two ReLU outputs that are both zero tie all the time."""
    from vltf_amd import ops
    rng = np.random.default_rng(2)
    a = rng.standard_normal((7, 5)).astype(np.float32)
    b, c = a.copy(), rng.standard_normal((7, 5)).astype(np.float32)
    b[::2] += 1.0                                       # rows 1, 3, 5 tie between a and b
    c[1] = a[1]                                         # row 1: a three-way tie where c is not below
    d = rng.standard_normal((7, 5)).astype(np.float32)
    for ins in ([a, b], [a, b, c]):
        want, _, _, _, cache = O.tensor_list_fusion(ins, "maximum", [5] * len(ins), [1] * len(ins), [1] * len(ins))
        wg = O.tensor_list_fusion_grad(cache, d)
        tin = [torch.from_numpy(x).to(DEV) for x in ins]
        out = torch.empty_like(tin[0])
        ops.fuse_n(tin, out, "maximum")
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        douts = [torch.empty_like(t) for t in tin]
        ops.fuse_n_grad(tin, torch.from_numpy(d).to(DEV), douts, "maximum")
        for got, w in zip(douts, wg):
            np.testing.assert_allclose(got.cpu().numpy(), w, rtol=1e-6, atol=0)
    ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    da, db = torch.empty_like(ta), torch.empty_like(ta)
    ops.max2_grad(ta, tb, torch.from_numpy(d).to(DEV), da, db)
    want, _, _, _, cache = O.tensor_list_fusion([a, b], "maximum", [5, 5], [1, 1], [1, 1])
    wa, wb = O.tensor_list_fusion_grad(cache, d)
    np.testing.assert_array_equal(da.cpu().numpy(), wa.astype(np.float32))
    np.testing.assert_array_equal(db.cpu().numpy(), wb.astype(np.float32))
    avg = torch.empty_like(ta)
    ops.fuse_n([ta, tb, torch.from_numpy(c).to(DEV)], avg, "avg")
    np.testing.assert_allclose(avg.cpu().numpy(), (a.astype(np.float64) + b + c) / 3, rtol=1e-6, atol=1e-7)


def _dp_worker(rank, world, port, q, name):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vltf_amd import dp
    from vltf_amd.graph import GraphEngine
    dp.init_from_env(backend="gloo")
    case = GC.CASES[name]()
    items = 4
    pipes, ds_all = GC.specs_and_datasets(case, items)
    raw, _ = GC.inputs(case, items)
    lo, hi = dp.shard_range(items, rank, world)
    _, ds = GC.specs_and_datasets(case, hi - lo)
    gar = dp.GradAllReduce()
    eng = GraphEngine(pipes, ds, case["V"], device=DEV, dp=gar)
    p = eng.init_params(seed=case["seed"], well_scaled=True)
    eng.load_params(p)
    gar.broadcast_params(eng.w)
    per_item = {t: d["cpv"] * d["fpc"] for t, d in case["data"].items()}              # input rows of one item, per dataset tag
    local = device_feeds({t: v[lo * per_item[t]:hi * per_item[t]] for t, v in raw.items()})
    rpi = eng.forward(local).shape[0] // (hi - lo)                                     # logit rows of one item
    onehot = O.labels_to_one_hot([[l] for l in np.random.default_rng(5).integers(0, case["V"], items * rpi)], case["V"])
    out = eng.train_step(local, torch.from_numpy(onehot[lo * rpi:hi * rpi]).to(DEV), lr=0.05, clip_norm=0.5, global_rows=items * rpi)
    got = eng.get_params()
    if rank == 0:
        ref = GraphEngine(pipes, ds_all, case["V"], device=DEV)
        ref.load_params(p)
        want_out = ref.train_step(device_feeds(raw), torch.from_numpy(onehot).to(DEV), lr=0.05, clip_norm=0.5)
        want = ref.get_params()
        err = {k: float(np.abs(got[k] - want[k]).max() / (np.abs(want[k] - p[k]).max() + 1e-12)) for k in want}
        worst = max(err, key=err.get)
        q.put(("ok", err[worst], worst, abs(out["grad_norm"] - want_out["grad_norm"]) / want_out["grad_norm"], gar.issued, len(eng.grad_chunks)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name,min_chunks", [("two_stream_avg", 5), ("dcnn_with_state", 3), ("fanout", 3), ("encdec_ibias_r1", 3)])
def test_two_rank_graph_step_equals_one_rank(name, min_chunks):
    """Data parallel over clips for multi-pipeline models: each pipeline's head chunk, then each tower's own chunk list through its
    offset in the shared flat buffer; two ranks (gloo, one GPU) end with the parameters of one rank stepping on all the items.
    `dcnn_with_state`, `fanout` and `encdec_ibias_r1` hold a `representation: fc` pipeline, whose fc_convert gradients round 3 exchanged
    before they were written (tests/test_graph_cpu.py pins the order; this is the end-to-end form).  (Not `encdec_ibias_r3`: at a
    clips-per-video ratio > 1 replicate_auxilliary_tensor tiles the WHOLE batch of pipeline-1 vectors, tf_util.py:182-192, so which vector
    meets which clip depends on the batch's composition -- a rank's shard pairs them as a smaller batch_size would in the reference, and
    no sharding reproduces the one-rank step there.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, name)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, "rank exited with %s" % pr.exitcode
    tag, worst, which, gn_err, issued, nchunks = q.get(timeout=10)
    assert tag == "ok" and issued == nchunks and nchunks >= min_chunks, (issued, nchunks)
    # parameter change relative to the largest change of that tensor; the two ranks sum half the rows each (another fp32 order):
    # 1.2e-3 measured on `fanout` (maximum fusion + late fusion); a gradient exchanged before it was written would be O(1) off
    assert worst < 3e-3 and gn_err < 1e-5, (worst, which, gn_err)


def test_two_stream_backward_under_allocator_churn():
    """The towers' backward runs its parameter gradients and the dgrad weight transposes on a second stream (engine._side_stream).
    A missing dependency there shows only when the streams drift apart -- tiny launches, fresh buffers -- and only now and then (one
    in a hundred steps, round 3: the first dgrad read weights the other stream had not finished transposing).  So: many FRESH
    engines of the two-tower graphs, the caching allocator perturbed between them with blocks of large values (stale memory must not
    look like zeros), every gradient of every step against the oracle."""
    from vltf_amd.graph import GraphEngine
    rng = np.random.default_rng(0)
    names = ["two_stream_concat", "two_stream_avg", "fused_frames_avg", "dcnn_with_state"]
    want, junk = {}, []
    for it in range(80):
        name = names[it % len(names)]
        case = GC.CASES[name]()
        pipes, ds = GC.specs_and_datasets(case)
        junk.append(torch.full((int(rng.integers(1, 64)) << 18,), 1e3, device=DEV))
        if len(junk) > 6:
            del junk[int(rng.integers(0, len(junk)))]
        eng = GraphEngine(pipes, ds, case["V"], device=DEV)
        p = eng.init_params(seed=case["seed"], well_scaled=True)
        raw, feeds = GC.inputs(case)
        if name not in want:
            want[name] = GC.expect(case, p, feeds)
        _, onehot, loss, grads, _ = want[name]
        eng.load_params(p)
        out = eng.train_step(device_feeds(raw), torch.from_numpy(onehot).to(DEV), lr=0.01, clip_norm=0.5)
        assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss)), (it, name)
        g = eng.get_grads()
        for k in p:
            scale = np.abs(grads[k]).max() + 1e-12
            np.testing.assert_allclose(g[k], grads[k], rtol=2e-3, atol=2e-4 * scale, err_msg="iteration %d, %s, grad %s" % (it, name, k))
        del eng
