"""Multi-pipeline models shared by tests/test_graph_cpu.py (host logic on the torch-CPU double) and tests/test_graph_gpu.py (the
device kernels): each case gives the YAML-shaped pipeline list (models/model.py:18-162, settings_.py:167-208), the datasets behind
its tags and synthetic inputs; `expect` evaluates the CPU oracle (oracle.lrcn_oracle.model_forward / model_backward) on them."""
import numpy as np

from oracle import lrcn_oracle as O

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)
SHAPE = (67, 67, 3)


def two_stream(fusion, H=6, layer="fc6"):
    """The two-stream LRCN: a dcnn feature pipeline per frame dataset, fused at equal rates into an LSTM classifier."""
    return dict(
        pipes=[("rgb", dict(input=["main"], representation="dcnn", frame_encoding_layer=layer, classifier=None)),
               ("flow", dict(input=["aux"], representation="dcnn", frame_encoding_layer=layer, classifier=None)),
               ("fuse", dict(input=["rgb", "flow"], input_fusion=fusion, representation="nop", classifier="lstm", lstm_params=[H, 1, "avg"]))],
        data={"main": dict(mode="video", fpc=3, cpv=1), "aux": dict(mode="video", fpc=3, cpv=1)}, V=7, items=2, seed=21)


def fanout():
    """Feature pipeline with early fusion consumed by two later pipelines, representation fc, classifier fc with late fusion, a
    pipeline nothing depends on, maximum fusion of two logits tensors, concat at ratio 1 into an fc classifier."""
    V = 6
    return dict(
        pipes=[("feat", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc7", classifier=None, frame_fusion=("early", "avg"))),
               ("a", dict(input=["feat"], representation="fc", fc_output_dim=V, classifier="fc")),
               ("perframe", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc8", classifier="fc", frame_fusion=("late", "last"))),
               ("unused", dict(input=["feat"], representation="fc", fc_output_dim=5, classifier=None)),
               ("ab", dict(input=["a", "perframe"], input_fusion="maximum", representation="nop", classifier=None)),
               ("out", dict(input=["ab", "feat"], input_fusion="concat", representation="nop", classifier="fc"))],
        data={"main": dict(mode="video", fpc=3, cpv=1)}, V=V, items=2, seed=27)


def dcnn_with_state():
    """A dcnn pipeline whose LSTM takes ANOTHER pipeline's output as its state vector (model.py:128-134): a vectors dataset through
    representation fc + early fusion gives one context vector per clip; 2 LSTM layers, input_state_fc (9 -> H)."""
    return dict(
        pipes=[("ctx", dict(input=["aux"], representation="fc", fc_output_dim=9, classifier=None, frame_fusion=("early", "avg"))),
               ("vis", dict(input=["main", "ctx"], representation="dcnn", frame_encoding_layer="fc6", classifier="lstm", lstm_params=[5, 2, "last"]))],
        data={"main": dict(mode="video", fpc=3, cpv=1), "aux": dict(mode="vectors", fpc=4, cpv=1, dim=6)}, V=7, items=2, seed=23)


def fused_frames(fusion="avg"):
    """ONE pipeline over the mean / maximum of two frame datasets (input_fusion applied to the placeholders, model.py:69-73)."""
    return dict(
        pipes=[("net", dict(input=["main", "aux"], input_fusion=fusion, representation="dcnn", frame_encoding_layer="fc6",
                            classifier="lstm", lstm_params=[6, 1, "avg"]))],
        data={"main": dict(mode="video", fpc=3, cpv=1), "aux": dict(mode="video", fpc=3, cpv=1)}, V=5, items=2, seed=24)


def encdec(input_fusion=None, ratio=1, fusion="reshape"):
    """BASELINE config 4's shape with the general engine: frames -> dcnn -> LSTM(state) => state / fused input of a word LSTM."""
    V = 7
    # ibias inserts pipeline 1's vector as one more time step: the word vectors must be as wide (tf_util.py:154-176); the input
    # fusion comes BEFORE the representation (model.py:69-96), so representation fc maps the fused T + 1 steps
    rep = dict(representation="fc", fc_output_dim=9) if input_fusion == "ibias" else dict(representation="nop")
    return dict(
        pipes=[("enc", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier="lstm", lstm_params=[6, 1, "state"])),
               ("dec", dict(input=["aux", "enc"], input_fusion=input_fusion, classifier="lstm", lstm_params=[8, 2, fusion], **rep))],
        data={"main": dict(mode="video", fpc=2, cpv=1), "aux": dict(mode="vectors", fpc=4, cpv=ratio, dim=V if input_fusion == "ibias" else 5)},
        V=V, items=2, seed=27)


CASES = {
    "two_stream_avg": lambda: two_stream("avg"), "two_stream_maximum": lambda: two_stream("maximum"),
    "two_stream_concat": lambda: two_stream("concat"), "fanout": fanout, "dcnn_with_state": dcnn_with_state,
    "fused_frames_avg": lambda: fused_frames("avg"), "fused_frames_maximum": lambda: fused_frames("maximum"),
    "encdec_state": lambda: encdec(), "encdec_concat_r2": lambda: encdec("concat", 2), "encdec_ibias_r3": lambda: encdec("ibias", 3),
    "encdec_ibias_r1": lambda: encdec("ibias", 1),
    "encdec_state_r2_avg": lambda: encdec(None, 2, "avg"),
}


def config4_full():
    """BASELINE config 4 at its own shapes (tools/bench_composed.py's job on 8 clips): AlexNet(fc6) + LSTM(256, state) over 16-frame
    227x227 clips => the state of a 256-unit LSTM over 21 word vectors (BOS + 20 tokens, 300-d), per-step logits over 1000 words."""
    return dict(
        pipes=[("enc", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier="lstm", lstm_params=[256, 1, "state"])),
               ("dec", dict(input=["aux", "enc"], representation="nop", classifier="lstm", lstm_params=[256, 1, "reshape"]))],
        data={"main": dict(mode="video", fpc=16, cpv=1), "aux": dict(mode="vectors", fpc=21, cpv=1, dim=300)},
        V=1000, items=8, seed=41, shape=(227, 227, 3))


def two_stream_full():
    """The two-stream LRCN at AlexNet's shapes: two dcnn towers over 4 clips x 16 frames of 227x227 each, fc6 features averaged into
    LSTM(256) -> 101 classes (128 frames through the conv stacks, as one rank's shard of the benchmark job)."""
    c = two_stream("avg", H=256)
    c["data"] = {"main": dict(mode="video", fpc=16, cpv=1), "aux": dict(mode="video", fpc=16, cpv=1)}
    c.update(V=101, items=4, seed=42, shape=(227, 227, 3))
    return c


# full-geometry cases whose oracle answers are committed (tests/golden/graph_full.npz, written by tests/golden/make_golden_graph.py)
FULL_CASES = {"c4_ws": config4_full, "ts_ws": two_stream_full}


def specs_and_datasets(case, items=None):
    """-> ([PipelineSpec], {tag: DatasetInfo}) for GraphEngine."""
    from vltf_amd.graph import DatasetInfo, PipelineSpec
    items = items or case["items"]
    pipes = [PipelineSpec(name=n, **{k: (tuple(v) if k in ("lstm_params", "frame_fusion") and v else v) for k, v in s.items()})
             for n, s in case["pipes"]]
    ds = {t: DatasetInfo(d["mode"], d["fpc"], d["cpv"], items * d["cpv"], image_shape=case.get("shape", SHAPE) if d["mode"] == "video" else None,
                         dim=d.get("dim")) for t, d in case["data"].items()}
    return pipes, ds


def inputs(case, items=None):
    """-> (raw {tag: uint8 frames | float32 vectors}, oracle feeds {tag: float arrays})."""
    rng = np.random.default_rng(case["seed"])
    items = items or case["items"]
    raw, feeds = {}, {}
    for t, d in sorted(case["data"].items()):
        rows = items * d["cpv"] * d["fpc"]
        if d["mode"] == "video":
            raw[t] = rng.integers(0, 256, (rows,) + tuple(case.get("shape", SHAPE)), dtype=np.uint8)
            feeds[t] = raw[t].astype(np.float32) - MEAN
        else:
            raw[t] = rng.standard_normal((rows, d["dim"])).astype(np.float32)
            feeds[t] = raw[t]
    return raw, feeds


def expect(case, p, feeds, lab_seed=0):
    """Oracle logits, labels for them, loss, gradients (only the pipelines the output depends on carry variables in p)."""
    ds = {t: dict(cpv=d["cpv"], fpc=d["fpc"]) for t, d in case["data"].items()}
    logits, cache = O.model_forward(p, case["pipes"], ds, feeds, case["V"])
    lab = np.random.default_rng(lab_seed).integers(0, case["V"], logits.shape[0])
    onehot = O.labels_to_one_hot([[l] for l in lab], case["V"])
    loss, dlogits = O.softmax_xent_mean(logits, onehot)
    return logits, onehot, loss, O.model_backward(p, cache, dlogits), cache
