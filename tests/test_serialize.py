"""The serializer (reference serialize.py) on CPU: clip / frame selection modes with the generation_error policies, the
clips-per-video check, shuffling and its side files, the `.size` sidecar, record contents and the read-back validator -- driven
through `serialize.main(<config.yml>)` over frame folders written here (lossless PNG so that a re-read is byte-identical)."""
import glob
import os
import random

import numpy as np
import pytest
import yaml

from vltf_amd import serialize as S
from vltf_amd import tfrecord
from vltf_amd.defs_ import defs

PIL = pytest.importorskip("PIL.Image")
RAW = (12, 16, 3)


def make_videos(folder, lengths, shape=(12, 16), seed=0, fmt="png"):
    rng = np.random.default_rng(seed)
    lines = []
    for v, n in enumerate(lengths):
        d = os.path.join(folder, "vid%02d" % v)
        os.makedirs(d)
        for k in range(n):
            PIL.fromarray(rng.integers(0, 256, shape + (3,), dtype=np.uint8)).save(os.path.join(d, "frame%04d.%s" % (k + 1, fmt)))
        lines.append("%s %d" % (d, v % 3) + (" 7" if v == 1 else ""))
    paths = os.path.join(folder, "videos.txt")
    open(paths, "w").write("\n".join(lines) + "\n")
    return paths


def config(folder, paths, mode, num, fpc, policy="abort", shuffle=False, out=None, raw=RAW, validate=True, seed=3, fmt="png"):
    cfg = {"serialize": {"output_folder": out, "path_prepend_folder": None, "input_files": [paths], "run_id": "ser",
                         "num_threads": 3, "num_items_per_thread": 4, "raw_image_shape": str(raw), "clip_offset_or_num": num,
                         "num_frames_per_clip": fpc, "clipframe_mode": "defs.clipframe_mode.%s" % mode,
                         "generation_error": "defs.generation_error.%s" % policy, "do_shuffle": shuffle, "do_serialize": True,
                         "do_validate": validate, "frame_format": fmt, "logging_level": "logging.INFO", "email_notify": None,
                         "seed": seed}}
    p = os.path.join(folder, "ser_%s.yml" % mode)
    yaml.safe_dump(cfg, open(p, "w"))
    return p


def records(path):
    return [tfrecord.parse_frame_example(p) for p in tfrecord.tf_record_iterator(path)]


def settings_for(raw=RAW):
    s = S.SerializationSettings()
    s.raw_image_shape = raw
    return s


def test_rand_clips_files_and_validation(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = str(tmp_path)
    paths = make_videos(folder, [9, 7, 12])
    written, errors = S.main(config(folder, paths, "rand_clips", 2, 3))
    assert errors == [False]
    size = open(paths + ".tfrecord.size").read()
    assert size == "items\t3\ntype\tvideo\ncpi\t[(3, 2)]\nfpc\t3\nlabelcount\t2\n"              # serialize.py:138-151
    listing = paths + ".2.cpv.3.fpc.rand_clips.cfm"                                                # serialize.py:806-814
    lines = open(listing).read().splitlines()
    recs = records(paths + ".tfrecord")
    assert len(recs) == len(lines) == 3 * 2 * 3
    st = settings_for()
    for (img, lab), line in zip(recs, lines):
        fp, *labs = line.split()
        assert np.array_equal(img, S.read_image(fp, st)) and list(lab) == [int(x) for x in labs]
    # clips are runs of consecutive frames of one video, labels those of the video (video 1 has two)
    for c in range(6):
        names = [l.split()[0] for l in lines[3 * c:3 * c + 3]]
        idx = [int(os.path.basename(n)[5:9]) for n in names]
        assert idx == list(range(idx[0], idx[0] + 3)) and len({os.path.dirname(n) for n in names}) == 1
    assert [len(l.split()) - 1 for l in lines[6:12]] == [2] * 6
    # the records are the reference's layout: BGR of the RGB file
    rgb = np.asarray(PIL.open(lines[0].split()[0]))
    assert np.array_equal(recs[0][0], rgb[:, :, ::-1])


def test_iterative_and_rand_frames(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = str(tmp_path)
    paths = make_videos(folder, [10, 10])
    S.main(config(folder, paths, "iterative", 1, 3))                      # starts fpc + offset = 4 apart: 0, 4 (7 would need frame 10)
    lines = open(paths + ".1.cpv.3.fpc.iterative.cfm").read().splitlines()
    idx = [int(os.path.basename(l.split()[0])[5:9]) for l in lines]
    assert idx == [1, 2, 3, 5, 6, 7] * 2
    assert "cpi\t[(2, 2)]" in open(paths + ".tfrecord.size").read()
    S.main(config(folder, paths, "rand_frames", 1, 4))                    # one clip of 4 distinct random frames per video
    lines = open(paths + ".4.fpc.rand_frames.cfm").read().splitlines()     # no cpv part in the name (serialize.py:806-807)
    assert len(lines) == 8
    for v in range(2):
        names = [l.split()[0] for l in lines[4 * v:4 * v + 4]]
        assert len(set(names)) == 4 and len({os.path.dirname(n) for n in names}) == 1
    assert "cpi\t[(2, 1)]" in open(paths + ".tfrecord.size").read() and len(records(paths + ".tfrecord")) == 8


def test_generation_error_policies(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = str(tmp_path)
    paths = make_videos(folder, [8, 2, 8])                                # video 1 has 2 frames, clips need 3
    with pytest.raises(Exception, match="cannot sustain a number of 3 fpc"):
        S.main(config(folder, paths, "rand_clips", 2, 3, policy="abort"))
    # compromise: the start frame is duplicated up to the clip length, in every requested clip
    written, errors = S.main(config(folder, paths, "rand_clips", 2, 3, policy="compromise"))
    assert errors == [False]
    lines = open(paths + ".2.cpv.3.fpc.rand_clips.cfm").read().splitlines()
    short = [os.path.basename(l.split()[0]) for l in lines if "vid01" in l]
    assert short == ["frame0001.png", "frame0001.png", "frame0002.png"] * 2
    assert len(records(paths + ".tfrecord")) == 18
    # report: nothing is serialized for that input, the offending folders are listed
    os.remove(paths + ".tfrecord")
    written, errors = S.main(config(folder, paths, "rand_clips", 2, 3, policy="report"))
    assert errors == [True] and written == [None] and not os.path.exists(paths + ".tfrecord")
    rep = glob.glob(os.path.join(folder, "generation_errors_files_ser_*"))
    # (listed once per logged problem: too short for a clip AND for the requested clip count, serialize.py:309-311,322-335)
    assert len(rep) == 1 and open(rep[0]).read().split() == [os.path.join(folder, "vid01")] * 2
    # too few clip starts: 4 clips of 3 from a 4-frame video (2 starts) -> compromise repeats starts
    st = S.SerializationSettings()
    st.num_frames_per_clip, st.clip_offset_or_num, st.generation_error = 3, 4, defs.generation_error.compromise
    random.seed(1)
    clips = S.get_random_clips(list(range(4)), st, "v")
    assert len(clips) == 4 and all(c in ([0, 1, 2], [1, 2, 3]) for c in clips) and len(st.generation_log) == 1
    st.generation_error = defs.generation_error.report
    assert S.get_random_clips(list(range(4)), st, "v") == []
    # rand_frames / iterative: compromise duplicates random frames
    st.generation_error, st.num_frames_per_clip = defs.generation_error.compromise, 5
    got = S.get_random_frames([0, 1, 2], st, "v")
    assert len(got) == 1 and len(got[0]) == 5 and set(got[0]) == {0, 1, 2}
    with pytest.raises(Exception, match="Erratic"):
        S.check_cpv_per_item([[["a"]], [["a"], ["b"]]], ["v0", "v1"], st.__class__())      # cpv 1 expected, v1 has 2


def test_shuffle_side_files_output_folder_and_resize(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = str(tmp_path)
    paths = make_videos(folder, [6, 6, 6, 6], shape=(20, 30))
    out = os.path.join(folder, "out")
    S.main(config(folder, paths, "rand_clips", 1, 2, shuffle=True, out=out, raw=(12, 16, 3)))
    base = os.path.join(out, "videos.txt")
    assert open(base + ".unshuffled").read() == open(paths).read()
    shuffled = open(base + ".shuffled").read().splitlines()
    assert sorted(shuffled) == sorted(open(paths).read().splitlines()) and os.path.exists(os.path.join(out, "ser_rand_clips.yml"))
    lines = open(base + ".1.cpv.2.fpc.rand_clips.cfm").read().splitlines()
    recs = records(base + ".tfrecord")
    order = [os.path.dirname(l.split()[0]) for l in lines[::2]]
    assert order == [l.split()[0] for l in shuffled]                      # records follow the shuffled video order
    st = settings_for((12, 16, 3))
    for (img, lab), line in zip(recs, lines):
        assert img.shape == (12, 16, 3) and np.array_equal(img, S.read_image(line.split()[0], st))      # imresize to raw_image_shape
    src = np.asarray(PIL.open(lines[0].split()[0]))[:, :, ::-1]
    want = np.asarray(PIL.fromarray(np.ascontiguousarray(src)).resize((16, 12), resample=PIL.BILINEAR))
    assert np.array_equal(recs[0][0], want)


def test_validator_detects_a_changed_record(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = str(tmp_path)
    paths = make_videos(folder, [5, 5])
    cfg = config(folder, paths, "rand_clips", 1, 2, validate=False)
    written, errors = S.main(cfg)
    st = S.SerializationSettings().initialize_from_file(cfg, configure_logging=False)
    S.validate(written, errors, st)                                       # intact file passes
    # swap the frame behind a serialized path: the stored record no longer equals a fresh read
    victim = open(paths + ".1.cpv.2.fpc.rand_clips.cfm").read().split()[0]
    PIL.fromarray(np.zeros((12, 16, 3), np.uint8)).save(victim)
    with pytest.raises(Exception, match="errors exist"):
        S.validate(written, errors, st)


def test_vectors_file(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = str(tmp_path)
    rng = np.random.default_rng(0)
    vecs = rng.standard_normal((6, 5)).astype(np.float32)
    inp = os.path.join(folder, "feats")
    with open(inp, "w") as f:
        for i, v in enumerate(vecs):
            f.write("%s %d\n" % (",".join(repr(float(x)) for x in v), i % 4))
    open(inp + ".ids", "w").write("\n".join("id%d x" % i for i in range(6)) + "\n")
    written, errors = S.main(config(folder, inp, "rand_clips", 1, 1, shuffle=True))
    assert errors == [False] and "type\tvectors" in open(inp + ".tfrecord.size").read()
    got = [tfrecord.parse_vector_example(p) for p in tfrecord.tf_record_iterator(inp + ".tfrecord")]
    (_, sidx), labels, ids, _, mode = written[0]
    assert mode == defs.input_mode.vectors and sorted(sidx) == list(range(6)) and ids == ["id%d" % i for i in sidx]
    for k, (v, lab) in enumerate(got):
        assert np.array_equal(v, vecs[sidx[k]]) and lab == [sidx[k] % 4]
