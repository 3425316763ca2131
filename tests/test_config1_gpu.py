"""BASELINE config 1 on the reference's own example frames (tests/golden/config1, made by tests/golden/make_config1_fixture.py from
examples/data/videos/videos.zip + paths.txt): `run_task.py` phase val of the single-frame AlexNet pipeline {dcnn @ fc8, classifier
fc, frame_fusion [late, avg]} over the committed TFRecord, per-video logits against the CPU oracle's committed answers (1e-3), with
imgproc center_crop and with imgproc resize -- the device-side scipy.misc.imresize (= PIL bilinear, dataset_.py:481-495), which
is also checked bit for bit on its own."""
import glob
import hashlib
import os
import pickle

import numpy as np
import pytest
import torch
import yaml

from oracle import lrcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "config1")
FPC, NCLS, RAW, WANT = 4, 6, (240, 320, 3), (227, 227, 3)
MEAN = [99.197148, 105.293620, 109.503945]


def fixture_frames():
    from vltf_amd import tfrecord
    return np.stack([tfrecord.parse_frame_example(p)[0] for p in tfrecord.tf_record_iterator(os.path.join(FIX, "frames.txt.tfrecord"))])


def test_fixture_is_the_reference_layout():
    from vltf_amd import tfrecord
    frames = fixture_frames()
    assert frames.shape == (8,) + RAW and frames.dtype == np.uint8
    assert tfrecord.read_size_file(os.path.join(FIX, "frames.txt.tfrecord.size")) == {"items": 2, "type": "video", "cpv": [1, 1], "fpc": FPC,
                                                                                       "labelcount": 1}
    assert open(os.path.join(FIX, "frames.txt")).read().split() == ["video_0000", "0", "video_0001", "5"]       # paths.txt labels 0 and 5
    assert frames.std() > 20                                                      # real image content, not noise or a constant


@pytest.mark.parametrize("src,dst", [((240, 320), (227, 227)), ((240, 320), (240, 320)), ((37, 53), (80, 90)), ((80, 90), (37, 53)),
                                     ((64, 64), (64, 31)), ((31, 64), (63, 64)), ((480, 640), (240, 320)), ((9, 7), (227, 227))])
def test_device_imresize_is_pil_bilinear_bit_for_bit(src, dst):
    from vltf_amd import ops
    rng = np.random.default_rng(src[0] * 131 + dst[1])
    if src == (240, 320):
        imgs = fixture_frames()
    else:
        imgs = np.concatenate([rng.integers(0, 256, (3,) + src + (3,), dtype=np.uint8),
                               (rng.integers(0, 2, (2,) + src + (3,)) * 255).astype(np.uint8)])
    got = ops.Resize(src[0], src[1], dst[0], dst[1])(torch.from_numpy(imgs).to(DEV)).cpu().numpy()
    want = np.stack([O.imresize_bilinear_u8(im, dst) for im in imgs])
    assert got.shape == want.shape and np.array_equal(got, want), "%d bytes differ" % int((got != want).sum())
    try:
        from PIL import Image
    except ImportError:
        return
    pil = np.stack([np.asarray(Image.fromarray(im).resize((dst[1], dst[0]), resample=Image.BILINEAR)) for im in imgs])
    assert np.array_equal(got, pil)
    if src == (240, 320) and dst == (227, 227):
        exp = np.load(os.path.join(FIX, "expected.npz"))
        assert [hashlib.sha256(g.tobytes()).hexdigest() for g in got] == list(exp["resize/sha256"])


def write_cfg(folder, imgproc, resume):
    cfg = {"run": {
        "resume_file": resume, "run_folder": os.path.join(folder, "run"), "run_id": "c1", "phase": "defs.phase.val",
        "data": {"ucf": {"data_path": os.path.join(FIX, "frames.txt"), "raw_image_shape": str(RAW), "image_shape": str(WANT),
                         "mean_image": MEAN, "data_format": "defs.data_format.tfrecord", "frame_format": "jpg",
                         "imgproc": ["defs.imgproc.%s" % imgproc, "defs.imgproc.sub_mean"], "phase": "defs.phase.val",
                         "tag": "defs.dataset_tag.main"}},
        "network": {"num_classes": NCLS, "pipelines": [{"alexnet": {
            "input": "defs.dataset_tag.main", "representation": "defs.representation.dcnn", "frame_encoding_layer": "fc8",
            "classifier": "defs.classifier.fc", "frame_fusion": ["defs.fusion_type.late", "defs.fusion_method.avg"]}}]},
        "val": {"batch_size": 2, "logits_save_interval": -1, "clip_fusion": ["defs.fusion_type.late", "defs.fusion_method.avg"]},
        "logging": {"save_freq_per_epoch": 1, "level": "logging.INFO", "print_tensors": False, "tensorboard_folder": "tb",
                    "email_notify": None}}}
    path = os.path.join(folder, "val_%s.yml" % imgproc)
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


@pytest.mark.parametrize("imgproc", ["center_crop", "resize"])
def test_config1_validation_on_reference_frames(tmp_path, monkeypatch, imgproc):
    monkeypatch.setenv("VLTF_CONV_MATH", "f32")
    from vltf_amd import run_task
    from vltf_amd.engine import NetConfig, init_params
    folder = str(tmp_path)
    cfg = NetConfig(image_shape=WANT, num_classes=NCLS, fpc=FPC, frame_encoding_layer="fc8", classifier="fc", frame_fusion=("late", "avg"))
    params = init_params(cfg, seed=11, well_scaled=True)                # the fixture's parameters
    os.makedirs(os.path.join(folder, "run", "checkpoints"))
    base = os.path.join(folder, "run", "checkpoints", "seeded.graph-0")
    np.savez(base + ".weights.npz", **params)
    with open(base + ".snap", "wb") as f:
        pickle.dump([0, 0, 0], f)
    acc = run_task.main(write_cfg(folder, imgproc, base))
    tot = glob.glob(os.path.join(folder, "run", "validation_logits_c1_val_resume_*.total"))
    assert len(tot) == 1
    with open(tot[0], "rb") as f:
        got = pickle.load(f)                                             # written by this run
    exp = np.load(os.path.join(FIX, "expected.npz"))
    want = exp[imgproc + "/video_logits"]
    assert got.shape == (2, NCLS) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3)
    assert acc == float(np.mean(want.argmax(1) == exp["labels"]))
    assert float(open(os.path.join(folder, "run", "accuracy_c1_val_resume")).read()) == acc
