"""Host logic of the workflow surface on CPU: YAML schema, dataset batching over TFRecords, LR table,
validation aggregation, checkpoint cadence.  No GPU: the engine is not constructed."""
import os

import numpy as np
import pytest
import yaml

from oracle import lrcn_oracle as O
from vltf_amd import serialize, settings_
from vltf_amd.defs_ import defs
from vltf_amd.train import precompute_learning_rates
from vltf_amd.val import Validation

MEAN = [99.197148, 105.293620, 109.503945]


def make_dataset(folder, name, nvid=5, fpc=3, cpv=(1, 2, 1, 1, 2), shape=(20, 24, 3), classes=4, seed=0):
    rng = np.random.default_rng(seed)
    videos = [rng.integers(0, 256, (c * fpc,) + shape, dtype=np.uint8) for c in cpv]
    labels = [int(rng.integers(0, classes)) for _ in range(nvid)]
    path = os.path.join(folder, name)
    serialize.write_video_dataset(path, videos, labels, fpc, list(cpv))
    return path, videos, labels


def config(folder, data_path, phase="train", **over):
    cfg = {"run": {
        "resume_file": None, "run_folder": os.path.join(folder, "run"), "run_id": "t", "phase": "defs.phase.%s" % phase,
        "data": {"d1": {"data_path": data_path, "raw_image_shape": "(20, 24, 3)", "image_shape": "(16, 16, 3)",
                        "mean_image": MEAN, "data_format": "defs.data_format.tfrecord", "frame_format": "jpg",
                        "imgproc": ["defs.imgproc.rand_crop", "defs.imgproc.rand_mirror", "defs.imgproc.sub_mean"] if phase == "train"
                        else ["defs.imgproc.center_crop", "defs.imgproc.sub_mean"],
                        "batch_item": "defs.batch_item.default", "phase": "defs.phase.%s" % phase, "tag": "defs.dataset_tag.main"}},
        "network": {"num_classes": 4, "pipelines": [{"lrcn": {
            "input": "defs.dataset_tag.main", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6",
            "classifier": "defs.classifier.lstm", "lstm_params": [8, 1, "defs.fusion_method.avg"]}}]},
        "train": {"batch_size": 2, "epochs": 2, "optimizer": "defs.optim.sgd", "base_lr": 0.05, "lr_mult": "None",
                  "lr_decay": ["defs.decay.exp", "defs.periodicity.drops", 3, 0.5], "clip_norm": 10, "dropout_keep_prob": 0.5},
        "val": {"batch_size": 2, "logits_save_interval": -1, "clip_fusion": ["defs.fusion_type.late", "defs.fusion_method.avg"]},
        "logging": {"save_freq_per_epoch": 1, "level": "logging.INFO", "print_tensors": False, "tensorboard_folder": "tb",
                    "email_notify": None}}}
    cfg["run"].update(over)
    path = os.path.join(folder, "cfg_%s.yml" % phase)
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def test_settings_and_dataset(tmp_path):
    folder = str(tmp_path)
    data_path, videos, labels = make_dataset(folder, "train.txt")
    s = settings_.Settings()
    feeder = s.initialize(config(folder, data_path))
    assert s.run_id == "t_train_scratch" and s.phase == defs.phase.train and s.num_classes == 4
    p = s.pipelines["lrcn"]
    assert p.lstm_params == [8, 1, "avg"] and p.frame_encoding_layer == "fc6" and p.input == ["main"]
    assert s.train.lr_decay == ["exp", "drops", 3, 0.5] and s.train.clip_norm == 10 and s.get_dropout() == 0.5
    assert os.path.exists(os.path.join(s.run_folder, "cfg_train.yml"))               # config copied into the run folder
    d = feeder.get_dataset_by_tag("main")[0]
    assert d.num_items == 5 and d.clips_per_video == [1, 2, 1, 1, 2] and d.num_frames_per_clip == 3
    assert d.batches == [2, 2, 1] and feeder.get_num_batches() == 3                   # dataset_.py:600-611
    assert d.crop_h == list(range(0, 3)) and d.crop_w == list(range(0, 7))            # range(0, raw-want-1)
    d.seed(1)
    seen = 0
    for b, vids in enumerate([(0, 1), (2, 3), (4,)]):
        assert feeder.loop()
        fdict, num_data, num_labels, padding = feeder.get_feed_dict()
        cpv = [d.clips_per_video[v] for v in vids]
        assert num_data == [sum(cpv) * 3] and num_labels == sum(cpv) and padding == 0
        want = np.concatenate([videos[v] for v in vids])
        np.testing.assert_array_equal(fdict["frames_u8"], want)
        onehot = fdict["labels"]
        assert onehot.dtype == np.int32 and onehot.shape == (sum(cpv), 4)
        assert [int(r.argmax()) for r in onehot] == [labels[v] for v in vids for _ in range(d.clips_per_video[v])]
        assert fdict["crop_y"].max() <= 2 and fdict["crop_x"].max() <= 6 and set(fdict["mirror"]) <= {0, 1}
        seen += 1
    assert not feeder.loop() and seen == 3
    feeder.rewind_datasets()
    assert feeder.loop() and d.offset == 0
    # resume fast-forward to batch 2 (videos 0..3 = 5 clips = 15 records)
    d.restore(2, 1)
    fdict, *_ = feeder.get_feed_dict()
    np.testing.assert_array_equal(fdict["frames_u8"], videos[4])
    # save cadence: ceil(3 / 1) = 3 batches, 2 saves
    feeder.compute_save_interval()
    assert (feeder.save_interval, feeder.num_saves) == (3, 2) and feeder.should_save(3) and not feeder.should_save(2)
    # LR table + schedule file (train.py:50-109): 6 steps, period ceil(6/3) = 2
    lrs = precompute_learning_rates(s, 3)
    assert lrs == O.precompute_learning_rates(0.05, ["exp", "drops", 3, 0.5], 3, 2)
    lines = open(os.path.join(s.run_folder, "t_train_scratch_lr_decay_schedule.txt")).read().splitlines()
    assert len(lines) == 6 and lines[0] == "Epoch 1/2, batch 1/3, lr 0.05000000" and lines[-1].endswith("lr 0.01250000")


def test_settings_rejects_bad_configs(tmp_path):
    folder = str(tmp_path)
    data_path, _, _ = make_dataset(folder, "d.txt")
    base = yaml.safe_load(open(config(folder, data_path)))
    bad = yaml.safe_load(yaml.safe_dump(base))
    bad["run"]["network"]["pipelines"][0]["lrcn"]["load_weights"] = "x.npy"          # the shipped example's stale key
    p = os.path.join(folder, "bad1.yml")
    yaml.safe_dump(bad, open(p, "w"))
    with pytest.raises(Exception, match="Undefined pipeline field"):
        settings_.Settings().initialize(p)
    bad = yaml.safe_load(yaml.safe_dump(base))
    bad["run"]["phase"] = "defs.phase.nope"
    yaml.safe_dump(bad, open(p, "w"))
    with pytest.raises(Exception):
        settings_.Settings().initialize(p)
    bad = yaml.safe_load(yaml.safe_dump(base))
    bad["run"]["data"]["d1"]["data_path"] = os.path.join(folder, "missing.txt")
    yaml.safe_dump(bad, open(p, "w"))
    with pytest.raises(Exception, match="does not exist"):
        settings_.Settings().initialize(p)


def test_validation_aggregation(tmp_path):
    folder = str(tmp_path)
    data_path, videos, labels = make_dataset(folder, "val.txt")
    s = settings_.Settings()
    feeder = s.initialize(config(folder, data_path, phase="val"))
    d = feeder.get_dataset_by_tag("main")[0]
    assert d.crop_mode == "center_crop" and (d.crop_h, d.crop_w) == (2, 4)            # floor((20-16)/2), floor((24-16)/2)
    val = Validation(s)
    rng = np.random.default_rng(3)
    all_logits = []
    while feeder.loop():
        fdict, *_ = feeder.get_feed_dict()
        assert not fdict["mirror"].any() and (fdict["crop_y"] == 2).all() and (fdict["crop_x"] == 4).all()
        logits = rng.standard_normal((len(fdict["labels"]), 4)).astype(np.float32)
        all_logits.append(logits)
        val.process_validation_logits(d, s, logits, fdict["labels"].astype(np.float32))
        val.save_validation_logits_chunk()
    val.save_validation_logits_chunk(save_all=True)
    want = O.clip_fusion_per_video(np.concatenate(all_logits), d.clips_per_video, "avg")
    np.testing.assert_allclose(val.item_logits, want, rtol=1e-6)
    acc = val.get_accuracy()
    assert acc == np.mean(want.argmax(1) == np.array(labels))
    files = [f for f in os.listdir(s.run_folder) if f.startswith("validation_logits_t_val_scratch_") and f.endswith(".total")]
    assert len(files) == 1
    import pickle
    with open(os.path.join(s.run_folder, files[0]), "rb") as f:
        stored = pickle.load(f)                                                       # written by this test
    assert stored.dtype == np.float32 and stored.shape == (5, 4)
