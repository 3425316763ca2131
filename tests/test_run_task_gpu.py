"""`run_task.py <config.yml>` end to end on the GPU: train from scratch over TFRecords, checkpoint, resume,
validate; the validation logits (pickled result file) must match the CPU oracle evaluated with the saved
weights on the same records (center crop + mean subtraction as dataset_.py:481-501)."""
import glob
import os
import pickle

import numpy as np
import pytest
import yaml

from oracle import lrcn_oracle as O
from tests.test_host_workflow import MEAN, make_dataset

pytestmark = pytest.mark.gpu
RAW, WANT = (80, 90, 3), (67, 67, 3)


def write_cfg(folder, name, data_path, phase, resume=None, epochs=2, optimizer="sgd", det=False, run="run"):
    cfg = {"run": {
        "resume_file": resume, "run_folder": os.path.join(folder, run), "run_id": "e2e", "phase": "defs.phase.%s" % phase,
        "data": {"d": {"data_path": data_path, "raw_image_shape": str(RAW), "image_shape": str(WANT), "mean_image": MEAN,
                       "data_format": "defs.data_format.tfrecord", "frame_format": "jpg",
                       "imgproc": ["defs.imgproc.rand_crop", "defs.imgproc.rand_mirror", "defs.imgproc.sub_mean"] if (phase == "train" and not det)
                       else ["defs.imgproc.center_crop", "defs.imgproc.sub_mean"],
                       "phase": "defs.phase.%s" % phase, "tag": "defs.dataset_tag.main"}},
        "network": {"num_classes": 4, "pipelines": [{"lrcn": {
            "input": "defs.dataset_tag.main", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6",
            "classifier": "defs.classifier.lstm", "lstm_params": [8, 1, "defs.fusion_method.avg"]}}]},
        "train": {"batch_size": 2, "epochs": epochs, "optimizer": "defs.optim.%s" % optimizer, "base_lr": 1e-4, "lr_mult": "None",
                  "lr_decay": ["defs.decay.exp", "defs.periodicity.interval", 2, 0.9], "clip_norm": 5, "dropout_keep_prob": 0.0},
        "val": {"batch_size": 2, "logits_save_interval": -1, "clip_fusion": ["defs.fusion_type.late", "defs.fusion_method.avg"]},
        "logging": {"save_freq_per_epoch": 1, "level": "logging.INFO", "print_tensors": False, "tensorboard_folder": "tb",
                    "email_notify": None}}}
    path = os.path.join(folder, name)
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


@pytest.mark.parametrize("prefetch,math", [("2", "f32"), ("0", "f32"), ("2", "bf16x3"), ("2", "bf16")])
def test_train_resume_validate(tmp_path, monkeypatch, prefetch, math):
    """prefetch 2: batches read and uploaded ahead of the loop by the feeder's background thread (the default);
    prefetch 0: the reference's synchronous feed.  Same checkpoints, logs, resume behaviour and validation results.
    math bf16x3: the whole workflow with the opt-in split-bf16 conv arithmetic (VLTF_CONV_MATH), same checks.
    math bf16: the whole workflow on the bf16 conv PATH (packed operands, csrc/conv_c8.hip): reduced precision by design, so the
    validation logits are held to 5e-2 of the oracle's on the trained weights instead of 1e-3."""
    monkeypatch.setenv("VLTF_PREFETCH", prefetch)
    monkeypatch.setenv("VLTF_CONV_MATH", math)
    from vltf_amd import run_task
    folder = str(tmp_path)
    train_path, _, _ = make_dataset(folder, "train.txt", shape=RAW, seed=1)
    val_path, vvideos, vlabels = make_dataset(folder, "val.txt", nvid=3, cpv=(2, 1, 2), shape=RAW, seed=2)
    run = os.path.join(folder, "run")

    run_task.main(write_cfg(folder, "train.yml", train_path, "train", epochs=2), seed=3)
    ck = sorted(glob.glob(os.path.join(run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
    snaps = sorted(glob.glob(os.path.join(run, "checkpoints", "*.snap")), key=os.path.getmtime)
    assert len(ck) == 2 and len(snaps) == 2                                           # one save per epoch (save_freq_per_epoch 1)
    with open(snaps[-1], "rb") as f:
        assert pickle.load(f) == [3, 1, 6]                                            # [batch_index, epoch_index, global_step]
    assert os.path.exists(os.path.join(run, "e2e_train_scratch_lr_decay_schedule.txt"))
    log = open(glob.glob(os.path.join(run, "log_e2e_train_scratch_*.log"))[0]).read()
    assert "Mode: [train], epoch:  1/ 2, batch    1 /    3" in log and "global step: 6" in log and "batch loss/nats" in log

    # resume the first checkpoint (end of epoch 1, gs 3) and finish: the step counter continues from the .snap
    first = ck[0][:-len(".weights.npz")]
    run_task.main(write_cfg(folder, "resume.yml", train_path, "train", resume=first, epochs=2), seed=99)
    log2 = open(glob.glob(os.path.join(run, "log_e2e_train_resume_*.log"))[0]).read()
    assert "Restored training snapshot of epoch 1" in log2 and "global step: 6" in log2 and "global step: 3," not in log2

    # validate the latest checkpoint
    acc = run_task.main(write_cfg(folder, "val.yml", val_path, "val", resume="latest"))
    assert float(open(os.path.join(run, "accuracy_e2e_val_resume")).read()) == acc
    tot = glob.glob(os.path.join(run, "validation_logits_e2e_val_resume_*.total"))
    assert len(tot) == 1
    with open(tot[0], "rb") as f:
        got = pickle.load(f)                                                          # written by this run
    assert got.dtype == np.float32 and got.shape == (3, 4)
    latest = max(glob.glob(os.path.join(run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
    with np.load(latest, allow_pickle=False) as z:
        params = {k: z[k] for k in z.files}
    cy, cx = O.center_crop_offsets(RAW, WANT)
    frames = np.concatenate(vvideos)
    x = np.stack([O.process_image(f, WANT, (cy, cx), MEAN) for f in frames])
    clip_logits, _ = O.lrcn_forward(params, x, 3)
    want = O.clip_fusion_per_video(clip_logits, [2, 1, 2], "avg")
    tol = 5e-2 if math == "bf16" else 1e-3
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    srt = np.sort(want, axis=1)
    if math != "bf16" or float((srt[:, -1] - srt[:, -2]).min()) > 2 * tol:          # (reduced precision may flip a near tie)
        assert acc == float(np.mean(want.argmax(1) == np.array(vlabels)))


def test_adam_resume_equals_uninterrupted(tmp_path, monkeypatch):
    """tf.train.Saver() keeps every global variable (feeder.py:201), i.e. also Adam's m / v slots and beta powers: a run resumed from
    the end-of-epoch-1 checkpoint must end with exactly the weights of the uninterrupted 2-epoch run (deterministic imgproc)."""
    monkeypatch.setenv("VLTF_PREFETCH", "0")
    monkeypatch.setenv("VLTF_CONV_MATH", "f32")
    from vltf_amd import run_task
    folder = str(tmp_path)
    train_path, _, _ = make_dataset(folder, "train.txt", shape=RAW, seed=1)

    def final_weights(run):
        ck = sorted(glob.glob(os.path.join(folder, run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
        with np.load(ck[-1], allow_pickle=False) as z:
            return ck, {k: z[k] for k in z.files}

    run_task.main(write_cfg(folder, "a.yml", train_path, "train", epochs=2, optimizer="adam", det=True, run="runA"), seed=3)
    ck, full = final_weights("runA")
    assert len(ck) == 2 and "__optimizer__/adam_m" in full and int(full["__optimizer__/step_count"][0]) == 6
    assert np.abs(full["__optimizer__/adam_v"]).max() > 0
    first = ck[0][:-len(".weights.npz")]
    run_task.main(write_cfg(folder, "b.yml", train_path, "train", resume=first, epochs=2, optimizer="adam", det=True, run="runA"), seed=77)
    _, resumed = final_weights("runA")
    assert int(resumed["__optimizer__/step_count"][0]) == 6
    for k in full:
        np.testing.assert_array_equal(resumed[k], full[k], err_msg=k)
    # a weights-only checkpoint (no optimizer state) still resumes, with a fresh optimizer and a warning
    bare = os.path.join(folder, "runA", "checkpoints", "bare.graph-3")
    np.savez(bare + ".weights.npz", **{k: v for k, v in np.load(first + ".weights.npz").items() if not k.startswith("__optimizer__/")})
    import shutil
    shutil.copy(first + ".snap", bare + ".snap")
    run_task.main(write_cfg(folder, "c.yml", train_path, "train", resume=bare, epochs=2, optimizer="adam", det=True, run="runA"), seed=77)
    log = "".join(open(f).read() for f in glob.glob(os.path.join(folder, "runA", "log_e2e_train_resume_*.log")))
    assert "no optimizer state" in log
