#!/usr/bin/env python3
"""BASELINE config 1 on the reference's OWN example frames: builds tests/golden/config1/ from
/root/reference/examples/data/videos/{videos.zip, paths.txt} (two UCF-101 frame folders, labels 0 and 5).

Runs in the build container only (it reads the reference's data files and decodes the JPEGs with Pillow, the library behind the
reference's scipy.misc.imread); what it commits is DATA:
  frames.txt / frames.txt.tfrecord / frames.txt.tfrecord.size   the first FPC frames of each video as the reference's serializer
        stores them (serialize.py:411-425: imread -> RGB to BGR -> imresize to raw_image_shape 240x320, a no-op for these
        320x240 frames; serialize.py:246-256 record layout), `iterative` clips, 1 clip per video, FPC frames per clip
  expected.npz   the CPU oracle's answers on those records for the single-frame AlexNet of config 1 (pipeline
        {dcnn @ fc8, classifier fc, frame_fusion [late, avg]}, num_classes 6, mean [99.197148, 105.293620, 109.503945]):
        per-frame fc8 logits and per-video logits with imgproc center_crop (227 from 240x320) and with imgproc resize
        (scipy imresize = PIL bilinear to 227x227), plus SHA-256 digests of the PIL-resized frames.
Parameters: vltf_amd.engine.init_params(cfg, seed=11, well_scaled=True) (regenerated from the seed by the tests).

usage (repo root):  python tests/golden/make_config1_fixture.py"""
import hashlib
import io
import os
import re
import sys
import zipfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lrcn_oracle as O  # noqa: E402

REF = "/root/reference/examples/data/videos"
OUT = os.path.join(ROOT, "tests", "golden", "config1")
FPC, NCLS, RAW, WANT = 4, 6, (240, 320, 3), (227, 227, 3)
MEAN = [99.197148, 105.293620, 109.503945]


def params():
    from vltf_amd.engine import NetConfig, init_params
    cfg = NetConfig(image_shape=WANT, num_classes=NCLS, fpc=FPC, frame_encoding_layer="fc8", classifier="fc", frame_fusion=("late", "avg"))
    return init_params(cfg, seed=11, well_scaled=True)


def main():
    from vltf_amd import serialize
    os.makedirs(OUT, exist_ok=True)
    z = zipfile.ZipFile(os.path.join(REF, "videos.zip"))
    videos, labels = [], []
    for line in open(os.path.join(REF, "paths.txt")):
        parts = line.split()
        if not parts:
            continue
        name = os.path.basename(parts[0])
        jpgs = sorted((n for n in z.namelist() if n.startswith(name + "/") and n.endswith(".jpg")),
                      key=lambda n: int(re.search(r"\.(\d+)\.jpg$", n).group(1)))
        frames = []
        for n in jpgs[:FPC]:
            rgb = np.asarray(Image.open(io.BytesIO(z.read(n))).convert("RGB"))            # imread
            bgr = rgb[:, :, ::-1]                                                           # serialize.py:421-422
            img = np.asarray(Image.fromarray(np.ascontiguousarray(bgr)).resize((RAW[1], RAW[0]), resample=Image.BILINEAR))   # :424-425
            frames.append(img)
        videos.append(np.stack(frames))
        labels.append([int(l) for l in parts[1:]])
    serialize.write_video_dataset(os.path.join(OUT, "frames.txt"), videos, labels, FPC, 1)
    p = params()
    allf = np.concatenate(videos)
    out = {"labels": np.array([l[0] for l in labels])}
    cy, cx = O.center_crop_offsets(RAW, WANT)
    xc = np.stack([O.process_image(f, WANT, (cy, cx), MEAN) for f in allf])
    resized = np.stack([np.asarray(Image.fromarray(f).resize((WANT[1], WANT[0]), resample=Image.BILINEAR)) for f in allf])
    assert all(np.array_equal(O.imresize_bilinear_u8(f, WANT[:2]), r) for f, r in zip(allf, resized))
    xr = np.stack([O.process_image(f, WANT, None, MEAN) for f in resized])
    for tag, x in (("center_crop", xc), ("resize", xr)):
        per_frame, _ = O.lrcn_forward(p, x, FPC, final_layer="fc8", classifier="fc", frame_fusion=None)
        per_video, _ = O.lrcn_forward(p, x, FPC, final_layer="fc8", classifier="fc", frame_fusion=("late", "avg"))
        out[tag + "/frame_logits"], out[tag + "/video_logits"] = per_frame, per_video
    out["resize/sha256"] = np.array([hashlib.sha256(r.tobytes()).hexdigest() for r in resized])
    out["resize/sample"] = resized[:, ::37, ::41, :].copy()
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **out)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
