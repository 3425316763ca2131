"""Multi-pipeline models through the workflow surface: `run_task.py <config.yml>` with THREE pipelines -- the two-stream LRCN of
models/model.py:18-162: a dcnn feature pipeline (classifier: None) over the `main` frame dataset, one over the `aux` frame dataset,
and a third pipeline that averages them (input_fusion avg, tf_util.py:142-143) into an LSTM classifier -- train from TFRecords,
checkpoint with pipeline-scoped variable names, validate; the pickled per-video logits must match the CPU oracle
(oracle.lrcn_oracle.model_forward) evaluated with the saved weights on the same records.  And the encoder-decoder pair with
`imgproc resize` on its frame dataset: validation must see the frames training saw (resized, not stored-size)."""
import glob
import os
import pickle

import numpy as np
import pytest
import yaml

from oracle import lrcn_oracle as O
from tests.test_host_workflow import MEAN, make_dataset

pytestmark = pytest.mark.gpu
RAW, WANT, V = (80, 90, 3), (67, 67, 3), 6
CPV = (1, 2, 1, 1, 2)
LOGGING = {"save_freq_per_epoch": 1, "level": "logging.INFO", "print_tensors": False, "tensorboard_folder": "tb", "email_notify": None}
TRAIN = {"batch_size": 2, "epochs": 2, "optimizer": "defs.optim.sgd", "base_lr": 1e-3, "lr_mult": "None", "lr_decay": "None",
         "clip_norm": 5, "dropout_keep_prob": 0.0}
VAL = {"batch_size": 2, "logits_save_interval": -1, "clip_fusion": ["defs.fusion_type.late", "defs.fusion_method.avg"]}
PIPES = [("rgb", dict(input=["main"], representation="dcnn", frame_encoding_layer="fc6", classifier=None)),
         ("flow", dict(input=["aux"], representation="dcnn", frame_encoding_layer="fc6", classifier=None)),
         ("fuse", dict(input=["rgb", "flow"], input_fusion="avg", representation="nop", classifier="lstm", lstm_params=[8, 1, "avg"]))]


def frames_entry(path, phase, tag, imgproc=("defs.imgproc.center_crop", "defs.imgproc.sub_mean")):
    return {"data_path": path, "raw_image_shape": str(RAW), "image_shape": str(WANT), "mean_image": MEAN,
            "data_format": "defs.data_format.tfrecord", "frame_format": "jpg", "imgproc": list(imgproc),
            "phase": "defs.phase.%s" % phase, "tag": "defs.dataset_tag.%s" % tag}


def write_two_stream_cfg(folder, name, rgb_path, flow_path, phase, resume=None):
    cfg = {"run": {
        "resume_file": resume, "run_folder": os.path.join(folder, "run"), "run_id": "ts", "phase": "defs.phase.%s" % phase,
        "data": {"rgb": frames_entry(rgb_path, phase, "main"), "flow": frames_entry(flow_path, phase, "aux")},
        "network": {"num_classes": V, "pipelines": [
            {"rgb": {"input": "defs.dataset_tag.main", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6"}},
            {"flow": {"input": "defs.dataset_tag.aux", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6"}},
            {"fuse": {"input": ["rgb", "flow"], "input_fusion": "defs.fusion_method.avg", "representation": "defs.representation.nop",
                      "classifier": "defs.classifier.lstm", "lstm_params": [8, 1, "defs.fusion_method.avg"]}}]},
        "train": TRAIN, "val": VAL, "logging": LOGGING}}
    path = os.path.join(folder, name)
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


@pytest.mark.parametrize("prefetch", ["2", "0"])
def test_two_stream_train_and_validate(tmp_path, monkeypatch, prefetch):
    """prefetch 2 (the default): BOTH frame datasets are read and uploaded ahead of the loop, each by its own BatchPrefetcher;
    prefetch 0: the reference's synchronous feed.  Same checkpoints, logs and validation results."""
    monkeypatch.setenv("VLTF_CONV_MATH", "f32")
    monkeypatch.setenv("VLTF_PREFETCH", prefetch)
    from vltf_amd import run_task
    folder = str(tmp_path)
    rpath, rgb, labels = make_dataset(folder, "rgb.txt", nvid=5, cpv=CPV, shape=RAW, classes=V, seed=7)
    fpath, flow, _ = make_dataset(folder, "flow.txt", nvid=5, cpv=CPV, shape=RAW, classes=V, seed=8)
    run = os.path.join(folder, "run")
    run_task.main(write_two_stream_cfg(folder, "train.yml", rpath, fpath, "train"), seed=3)
    ck = sorted(glob.glob(os.path.join(run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
    assert len(ck) == 2
    with np.load(ck[-1], allow_pickle=False) as z:
        params = {k: z[k] for k in z.files if not k.startswith("__optimizer__/")}
    assert {"rgb/dcnn/conv1W", "flow/dcnn/conv1W", "fuse/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel", "fuse/output_fc_w"} <= set(params)
    assert not np.array_equal(params["rgb/dcnn/conv1W"], params["flow/dcnn/conv1W"])
    log = open(glob.glob(os.path.join(run, "log_ts_train_scratch_*.log"))[0]).read()
    assert "global step: 6" in log and "batch loss/nats" in log

    acc = run_task.main(write_two_stream_cfg(folder, "val.yml", rpath, fpath, "val", resume="latest"))
    tot = glob.glob(os.path.join(run, "validation_logits_ts_val_resume_*.total"))
    assert len(tot) == 1
    with open(tot[0], "rb") as f:
        got = pickle.load(f)                                                          # written by this run
    assert got.shape == (5, V)
    cy, cx = O.center_crop_offsets(RAW, WANT)
    feeds = {t: np.stack([O.process_image(f, WANT, (cy, cx), MEAN) for f in np.concatenate(v)]) for t, v in (("main", rgb), ("aux", flow))}
    clip_logits, _ = O.model_forward(params, PIPES, {"main": dict(cpv=1, fpc=3), "aux": dict(cpv=1, fpc=3)}, feeds, V)
    want = O.clip_fusion_per_video(clip_logits, list(CPV), "avg")
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3)
    assert acc == float(np.mean(want.argmax(1) == np.array(labels)))


def test_encoder_decoder_validation_resizes_frames_like_training(tmp_path, monkeypatch):
    """imgproc `resize` (dataset_.py:481-495: imresize to image_shape, no crop) on the frame dataset of the two-pipeline model: the
    validation pass hands the resize chain to the engine exactly as the train step does, so the logits it saves are the oracle's on
    the RESIZED frames (with the stored 80 x 90 frames the 67 x 67 network would not even accept them)."""
    monkeypatch.setenv("VLTF_CONV_MATH", "f32")
    from vltf_amd import run_task, serialize
    folder = str(tmp_path)
    TW, E = 4, 5
    fpath, videos, _ = make_dataset(folder, "frames.txt", nvid=4, cpv=(1, 1, 1, 1), shape=RAW, classes=V, seed=9)
    rng = np.random.default_rng(10)
    seqs = [rng.standard_normal((TW, E)).astype(np.float32) for _ in range(4)]
    wl = [[[int(rng.integers(0, V))] for _ in range(TW)] for _ in range(4)]
    wpath = os.path.join(folder, "words.txt")
    serialize.write_vector_dataset(wpath, seqs, wl, TW, 1)
    run = os.path.join(folder, "run")

    def cfg(name, phase, resume=None):
        c = {"run": {
            "resume_file": resume, "run_folder": run, "run_id": "rs", "phase": "defs.phase.%s" % phase,
            "data": {"frames": frames_entry(fpath, phase, "aux", ("defs.imgproc.resize", "defs.imgproc.sub_mean")),
                     "words": {"data_path": wpath, "data_format": "defs.data_format.tfrecord", "phase": "defs.phase.%s" % phase,
                               "tag": "defs.dataset_tag.main"}},
            "network": {"num_classes": V, "pipelines": [
                {"enc": {"input": "defs.dataset_tag.aux", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6",
                         "classifier": "defs.classifier.lstm", "lstm_params": [8, 1, "defs.fusion_method.state"]}},
                {"dec": {"input": ["defs.dataset_tag.main", "enc"], "representation": "defs.representation.nop",
                         "classifier": "defs.classifier.lstm", "lstm_params": [10, 1, "defs.fusion_method.reshape"]}}]},
            "train": dict(TRAIN, epochs=1), "val": VAL, "logging": LOGGING}}
        path = os.path.join(folder, name)
        with open(path, "w") as f:
            yaml.safe_dump(c, f)
        return path
    run_task.main(cfg("train.yml", "train"), seed=4)
    ck = sorted(glob.glob(os.path.join(run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
    with np.load(ck[-1], allow_pickle=False) as z:
        params = {k: z[k] for k in z.files if not k.startswith("__optimizer__/")}
    run_task.main(cfg("val.yml", "val", resume="latest"))
    tot = glob.glob(os.path.join(run, "validation_logits_rs_val_resume_*.total"))
    with open(tot[0], "rb") as f:
        got = pickle.load(f)                                                          # written by this run
    x = np.stack([O.process_image(O.imresize_bilinear_u8(f, WANT[:2]), WANT, None, MEAN) for f in np.concatenate(videos)])
    want, _ = O.encdec_forward(params, x, np.concatenate(seqs), 3, TW, dict(layer="fc6", layers=1), dict(layers=1, fusion="reshape"), V)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3)


def _dp_worker(rank, world, port, cfg_path, val_cfg):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      VLTF_DIST_BACKEND="gloo", VLTF_CONV_MATH="f32")
    from vltf_amd import run_task
    run_task.main(cfg_path, seed=3, device="cuda:0")
    run_task.main(val_cfg, device="cuda:0")


def test_two_rank_two_stream_run_task_equals_one_rank(tmp_path, monkeypatch):
    """The three-pipeline model data parallel through the workflow: two ranks (gloo, one GPU) each take their videos of every global
    batch of BOTH datasets; 5 videos at a global batch of 4 -> the last batch has one video, so rank 1's shard of it is EMPTY
    (GraphEngine.train_step_empty joins the exchange with zeros).  Rank 0's checkpoint and the sharded validation must equal the
    one-process run's."""
    import socket
    import torch.multiprocessing as mp
    monkeypatch.setenv("VLTF_CONV_MATH", "f32")
    from vltf_amd import run_task
    runs = {}
    for tag in ("one", "two"):
        folder = str(tmp_path / tag)
        os.makedirs(folder)
        rpath, _, _ = make_dataset(folder, "rgb.txt", nvid=5, cpv=CPV, shape=RAW, classes=V, seed=7)
        fpath, _, _ = make_dataset(folder, "flow.txt", nvid=5, cpv=CPV, shape=RAW, classes=V, seed=8)
        cfgs = []
        for name, phase, resume in (("train.yml", "train", None), ("val.yml", "val", "latest")):
            path = write_two_stream_cfg(folder, name, rpath, fpath, phase, resume)
            with open(path) as f:
                c = yaml.safe_load(f)
            c["run"]["train"]["batch_size"] = 4
            c["run"]["val"]["batch_size"] = 4
            with open(path, "w") as f:
                yaml.safe_dump(c, f)
            cfgs.append(path)
        runs[tag] = (folder, cfgs)
    folder, (tcfg, vcfg) = runs["one"]
    run_task.main(tcfg, seed=3)
    acc1 = run_task.main(vcfg)
    folder2, (tcfg2, vcfg2) = runs["two"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, tcfg2, vcfg2)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
    codes = [pr.exitcode for pr in procs]
    for pr in procs:
        if pr.is_alive():                         # a rank whose peer died waits in the rendezvous: do not leave it behind
            pr.terminate()
    assert codes == [0, 0], "ranks exited with %s" % codes

    def weights(folder):
        ck = sorted(glob.glob(os.path.join(folder, "run", "checkpoints", "*.weights.npz")), key=os.path.getmtime)
        with np.load(ck[-1], allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    want, got = weights(folder), weights(folder2)
    assert set(want) == set(got)
    for k in want:
        # (the LSTM bias starts at 0 and moves by ~1e-6 under these saturated gates: an absolute floor for summation-order noise;
        # two ranks sum every gradient in another order than one, and a few steps of this small net amplify that to ~3e-5 of a weight)
        assert np.abs(got[k] - want[k]).max() < 1e-4 * np.abs(want[k]).max() + 2e-8, k
    t1 = glob.glob(os.path.join(folder, "run", "validation_logits_*.total"))
    t2 = glob.glob(os.path.join(folder2, "run", "validation_logits_*.total"))
    assert len(t1) == 1 and len(t2) == 1
    with open(t1[0], "rb") as f1, open(t2[0], "rb") as f2:
        l1, l2 = pickle.load(f1), pickle.load(f2)                     # files these runs wrote
    assert l1.shape == l2.shape == (5, V)
    np.testing.assert_allclose(l2, l1, rtol=1e-4, atol=1e-4)
    assert float(open(glob.glob(os.path.join(folder2, "run", "accuracy_*"))[0]).read()) == acc1
