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


def test_vectors_dataset_and_two_pipeline_settings(tmp_path):
    """`vectors` input mode (serialize.py:258-266, dataset_.py:137-168,709-714): float32 vector records with per-record labels,
    read back batch by batch (per-clip labels = the first record's, per-record targets beside them), sharded, fast-forwarded;
    and the YAML of a two-pipeline model (settings_.py:167-208: a pipeline name as the input of a later pipeline)."""
    from vltf_amd import tfrecord
    from vltf_amd.dataset_ import Dataset
    folder = str(tmp_path)
    rng = np.random.default_rng(0)
    T, E, C = 4, 5, 7
    seqs = [rng.standard_normal((T, E)).astype(np.float32) for _ in range(5)]
    labels = [[[int(rng.integers(0, C))] for _ in range(T)] for _ in range(5)]
    path = os.path.join(folder, "w.txt")
    serialize.write_vector_dataset(path, seqs, labels, T, 1)
    assert tfrecord.read_size_file(path + ".tfrecord.size") == {"items": 5, "type": "vectors", "cpv": [1] * 5, "fpc": T, "labelcount": 1}
    d = Dataset()
    d.initialize("w", path, None, None, None, [], None, defs.data_format.tfrecord, None, defs.batch_item.default, C, "main", 1)
    d.calculate_batches(2, defs.input_mode.video)                       # the Feeder passes `video`; the size file decides
    assert d.input_mode == defs.input_mode.vectors and d.vector_dim() == E and len(d.batches) == 3
    vecs, cy, cx, mir, onehot = d.get_next_batch()
    assert cy is None and vecs.shape == (2 * T, E) and np.array_equal(vecs, np.concatenate(seqs[:2]))
    assert onehot.shape == (2, C) and onehot[0, labels[0][0][0]] == 1 and onehot[1, labels[1][0][0]] == 1
    want = O.labels_to_one_hot([l for item in labels[:2] for l in item], C)
    assert np.array_equal(d.record_onehot, want)
    d.get_next_batch()
    vecs, _, _, _, onehot = d.get_next_batch()                          # ragged last batch
    assert vecs.shape == (T, E) and np.array_equal(vecs, seqs[4])
    d.restore(1, 0)                                                     # resume at batch index 1
    assert np.array_equal(d.get_next_batch()[0], np.concatenate(seqs[2:4]))
    d.rewind()
    d.set_shard(1, 2)                                                   # rank 1 of 2 reads the second item of each batch
    assert np.array_equal(d.get_next_batch()[0], seqs[1]) and d.global_clips == 2
    assert np.array_equal(d.get_next_batch()[0], seqs[3])

    fpath, _, _ = make_dataset(folder, "f.txt")
    cfg = yaml.safe_load(open(config(folder, fpath)))
    cfg["run"]["data"]["d1"]["tag"] = "defs.dataset_tag.aux"
    cfg["run"]["data"]["w"] = {"data_path": path, "data_format": "defs.data_format.tfrecord", "phase": "defs.phase.train",
                               "tag": "defs.dataset_tag.main"}
    cfg["run"]["network"]["num_classes"] = C
    cfg["run"]["network"]["pipelines"] = [
        {"enc": {"input": "defs.dataset_tag.aux", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6",
                 "classifier": "defs.classifier.lstm", "lstm_params": [8, 1, "defs.fusion_method.state"]}},
        {"dec": {"input": ["defs.dataset_tag.main", "enc"], "representation": "defs.representation.fc", "fc_output_dim": 6,
                 "classifier": "defs.classifier.lstm", "lstm_params": [10, 2, "defs.fusion_method.reshape"],
                 "input_fusion": "defs.fusion_method.ibias"}}]
    p2 = os.path.join(folder, "two.yml")
    yaml.safe_dump(cfg, open(p2, "w"))
    s = settings_.Settings()
    s.initialize(p2)
    assert s.pipeline_names == ["enc", "dec"] and s.pipelines["dec"].input == ["main", "enc"]
    assert s.pipelines["dec"].input_fusion == "ibias" and s.pipelines["dec"].fc_output_dim == 6
    assert s.pipelines["dec"].lstm_params == [10, 2, "reshape"]
    cfg["run"]["network"]["pipelines"][1]["dec"]["input"] = ["defs.dataset_tag.main", "nosuch"]
    yaml.safe_dump(cfg, open(p2, "w"))
    with pytest.raises(Exception, match="not a dataset tag"):
        settings_.Settings().initialize(p2)


def test_weights_file_dict_npy_is_read_without_running_code(tmp_path):
    """alexnet.py:50-52: the pickled {layer: [W, b]} dict of bvlc_alexnet.npy.  Read by an arrays-only unpickler: the arrays come
    back bit for bit under their TF variable names, a pickle that names anything else is refused before it can run."""
    import pickle
    from vltf_amd.run_task import load_weights_file
    rng = np.random.default_rng(0)
    net = {"conv1": [rng.standard_normal((11, 11, 3, 96)).astype(np.float32), rng.standard_normal(96).astype(np.float32)],
           "fc6": [rng.standard_normal((16, 8)).astype(np.float32), np.zeros(8, np.float32)]}
    path = str(tmp_path / "net.npy")
    np.save(path, np.array(net, dtype=object), allow_pickle=True)               # a file this test wrote, in the reference's layout
    got = load_weights_file(path)
    assert sorted(got) == ["dcnn/conv1W", "dcnn/conv1b", "dcnn/fc6W", "dcnn/fc6b"]
    assert np.array_equal(got["dcnn/conv1W"], net["conv1"][0]) and np.array_equal(got["dcnn/fc6b"], net["fc6"][1])
    # python-2 style pickle (protocol 2, the age of bvlc_alexnet.npy) loads too
    path2 = str(tmp_path / "net2.npy")
    with open(path2, "wb") as f:
        np.lib.format.write_array_header_1_0(f, {"descr": "|O", "fortran_order": False, "shape": ()})
        pickle.dump(net, f, protocol=2)
    assert np.array_equal(load_weights_file(path2)["dcnn/conv1b"], net["conv1"][1])
    # anything but array reconstruction is refused
    evil = str(tmp_path / "evil.npy")
    marker = str(tmp_path / "pwned")

    class Evil:
        def __reduce__(self):
            import os as _os
            return (_os.system, ("touch %s" % marker,))
    with open(evil, "wb") as f:
        np.lib.format.write_array_header_1_0(f, {"descr": "|O", "fortran_order": False, "shape": ()})
        pickle.dump({"conv1": [Evil(), Evil()]}, f, protocol=2)
    with pytest.raises(Exception, match="refused"):
        load_weights_file(evil)
    assert not os.path.exists(marker)
    # npz keyed by TF names, no pickle at all
    npz = str(tmp_path / "w.npz")
    np.savez(npz, **{"dcnn/conv1W": net["conv1"][0]})
    assert np.array_equal(load_weights_file(npz)["dcnn/conv1W"], net["conv1"][0])


def test_graph_config_from_yaml(tmp_path):
    """run_task.graph_config: a three-pipeline `network:` block (two dcnn feature pipelines on main / aux fused into an LSTM) ->
    PipelineSpec list + DatasetInfo per tag, exactly what Model.__init__ walks (model.py:157-162); `input_shape` that contradicts
    the dataset is refused (model.py:47-54).  Host logic only: no engine is built."""
    from vltf_amd import run_task
    folder = str(tmp_path)
    p1, _, _ = make_dataset(folder, "rgb.txt", cpv=(1, 2, 1, 1, 2))
    p2, _, _ = make_dataset(folder, "flow.txt", cpv=(1, 2, 1, 1, 2), seed=1)
    path = config(folder, p1)
    with open(path) as f:
        cfg = yaml.safe_load(f)
    d1 = cfg["run"]["data"]["d1"]
    cfg["run"]["data"]["d2"] = dict(d1, data_path=p2, tag="defs.dataset_tag.aux")
    cfg["run"]["network"]["pipelines"] = [
        {"rgb": {"input": "defs.dataset_tag.main", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc6"}},
        {"flow": {"input": "defs.dataset_tag.aux", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc7",
                  "input_shape": "(16, 16, 3)"}},
        {"fuse": {"input": ["rgb", "flow"], "input_fusion": "defs.fusion_method.maximum", "representation": "defs.representation.nop",
                  "classifier": "defs.classifier.lstm", "lstm_params": [8, 2, "defs.fusion_method.last"]}}]
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    s = settings_.Settings()
    feeder = s.initialize(path)
    specs, infos, by_tag = run_task.graph_config(s, feeder, batch=2)
    assert [sp.name for sp in specs] == ["rgb", "flow", "fuse"] and specs[2].input == ["rgb", "flow"]
    assert specs[0].classifier is None and specs[1].frame_encoding_layer == "fc7" and specs[2].input_fusion == "maximum"
    assert specs[2].lstm_params == (8, 2, "last") and specs[2].representation == "nop"
    assert set(infos) == {"main", "aux"} and infos["main"].mode == "video" and infos["main"].fpc == 3 and infos["main"].cpv == 1
    assert infos["aux"].max_clips == 2 * 2 and tuple(infos["aux"].image_shape) == (16, 16, 3) and set(by_tag) == {"main", "aux"}
    cfg["run"]["network"]["pipelines"][1]["flow"]["input_shape"] = "(20, 24, 3)"
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    s2 = settings_.Settings()
    feeder2 = s2.initialize(path)
    with pytest.raises(Exception, match="input_shape"):
        run_task.graph_config(s2, feeder2, batch=2)
