"""BASELINE config 4 through the workflow surface: `run_task.py <config.yml>` with TWO pipelines (a dcnn + LSTM(state) encoder over
a frame dataset feeding an LSTM decoder over a `vectors` dataset as its state vector, per-step logits, word-level targets) -- train
from TFRecords, checkpoint with pipeline-scoped variable names, validate; the pickled per-step logits must match the CPU oracle
(oracle.lrcn_oracle.encdec_forward) evaluated with the saved weights on the same records."""
import glob
import os
import pickle

import numpy as np
import pytest
import yaml

from oracle import lrcn_oracle as O
from tests.test_host_workflow import MEAN, make_dataset

pytestmark = pytest.mark.gpu
RAW, WANT, V, E, TW = (80, 90, 3), (67, 67, 3), 9, 6, 5


def make_words(folder, name, nitems, seed):
    from vltf_amd import serialize
    rng = np.random.default_rng(seed)
    seqs = [rng.standard_normal((TW, E)).astype(np.float32) for _ in range(nitems)]
    labels = [[[int(rng.integers(0, V))] for _ in range(TW)] for _ in range(nitems)]        # a target per record (next word id)
    path = os.path.join(folder, name)
    serialize.write_vector_dataset(path, seqs, labels, TW, 1)
    return path, seqs, labels


def write_cfg(folder, name, frames_path, words_path, phase, resume=None):
    imgproc = ["defs.imgproc.center_crop", "defs.imgproc.sub_mean"]
    cfg = {"run": {
        "resume_file": resume, "run_folder": os.path.join(folder, "run"), "run_id": "c4", "phase": "defs.phase.%s" % phase,
        "data": {
            "frames": {"data_path": frames_path, "raw_image_shape": str(RAW), "image_shape": str(WANT), "mean_image": MEAN,
                       "data_format": "defs.data_format.tfrecord", "frame_format": "jpg", "imgproc": imgproc,
                       "phase": "defs.phase.%s" % phase, "tag": "defs.dataset_tag.aux"},
            "words": {"data_path": words_path, "data_format": "defs.data_format.tfrecord", "phase": "defs.phase.%s" % phase,
                      "tag": "defs.dataset_tag.main"}},
        "network": {"num_classes": V, "pipelines": [
            {"enc": {"input": "defs.dataset_tag.aux", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6",
                     "classifier": "defs.classifier.lstm", "lstm_params": [8, 1, "defs.fusion_method.state"]}},
            {"dec": {"input": ["defs.dataset_tag.main", "enc"], "representation": "defs.representation.nop",
                     "classifier": "defs.classifier.lstm", "lstm_params": [10, 1, "defs.fusion_method.reshape"]}}]},
        "train": {"batch_size": 2, "epochs": 2, "optimizer": "defs.optim.sgd", "base_lr": 1e-3, "lr_mult": "None",
                  "lr_decay": "None", "clip_norm": 5, "dropout_keep_prob": 0.0},
        "val": {"batch_size": 2, "logits_save_interval": -1, "clip_fusion": ["defs.fusion_type.late", "defs.fusion_method.avg"]},
        "logging": {"save_freq_per_epoch": 1, "level": "logging.INFO", "print_tensors": False, "tensorboard_folder": "tb",
                    "email_notify": None}}}
    path = os.path.join(folder, name)
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def test_two_pipeline_train_and_validate(tmp_path, monkeypatch):
    monkeypatch.setenv("VLTF_CONV_MATH", "f32")
    from vltf_amd import run_task
    folder = str(tmp_path)
    fpath, videos, _ = make_dataset(folder, "frames.txt", nvid=5, cpv=(1, 1, 1, 1, 1), shape=RAW, seed=4)
    wpath, seqs, labels = make_words(folder, "words.txt", 5, seed=5)
    run = os.path.join(folder, "run")
    run_task.main(write_cfg(folder, "train.yml", fpath, wpath, "train"), seed=3)
    ck = sorted(glob.glob(os.path.join(run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
    assert len(ck) == 2
    with np.load(ck[-1], allow_pickle=False) as z:
        params = {k: z[k] for k in z.files if not k.startswith("__optimizer__/")}
    assert "enc/dcnn/conv1W" in params and "dec/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel" in params
    assert "dec/input_state_fc_w" in params and "enc/fc_convert_w" not in params or params["enc/fc_convert_w"].shape == (8, V)
    log = open(glob.glob(os.path.join(run, "log_c4_train_scratch_*.log"))[0]).read()
    assert "global step: 6" in log and "batch loss/nats" in log

    acc = run_task.main(write_cfg(folder, "val.yml", fpath, wpath, "val", resume="latest"))
    tot = glob.glob(os.path.join(run, "validation_logits_c4_val_resume_*.total"))
    assert len(tot) == 1
    with open(tot[0], "rb") as f:
        got = pickle.load(f)                                                          # written by this run
    assert got.shape == (5 * TW, V)
    cy, cx = O.center_crop_offsets(RAW, WANT)
    x = np.stack([O.process_image(f, WANT, (cy, cx), MEAN) for f in np.concatenate(videos)])
    words = np.concatenate(seqs)
    want, _ = O.encdec_forward(params, x, words, 3, TW, dict(layer="fc6", layers=1), dict(layers=1, fusion="reshape"), V)
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3)
    flat = np.array([l[0] for item in labels for l in item])
    assert acc == float(np.mean(want.argmax(1) == flat))
