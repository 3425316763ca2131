"""Writer side of the input format (serialize.py:138-151,246-256,269-378): frames -> TFRecord + .size.
Only what the path's tests and examples need: videos given as arrays / frame folders, clip sampling modes
`iterative` and `rand_clips`, one label set per video.  The full serialize CLI is out of scope (SURVEY f-1)."""
import os
import random

import numpy as np

from . import tfrecord
from .defs_ import defs
from .utils_ import error, info


def generate_clips(num_frames, fpc, clip_offset_or_num, mode, rng=random):
    """-> list of frame-index lists.  iterative (serialize.py:322-346): consecutive clips `clip_offset_or_num`
    frames apart; rand_clips (298-320): that many clips at random starts."""
    if num_frames < fpc:
        return []
    if mode == defs.clipframe_mode.iterative:
        starts = list(range(0, num_frames - fpc + 1, max(1, clip_offset_or_num)))
    elif mode == defs.clipframe_mode.rand_clips:
        possible = list(range(0, num_frames - fpc + 1))
        starts = sorted(rng.sample(possible, min(clip_offset_or_num, len(possible))))
    else:
        error("clipframe mode [%s] is not built" % mode)
    return [list(range(s, s + fpc)) for s in starts]


def write_video_dataset(paths_file, videos, labels, fpc, clips_per_video):
    """videos: list of uint8 arrays [frames, H, W, 3] (BGR); writes <paths_file>, .tfrecord and .tfrecord.size with
    `clips_per_video[i]` clips of `fpc` consecutive frames each (clip j starts at frame j*fpc)."""
    cpv = clips_per_video if isinstance(clips_per_video, (list, tuple)) else [clips_per_video] * len(videos)
    with open(paths_file, "w") as f:
        for i, lab in enumerate(labels):
            lab = lab if isinstance(lab, (list, tuple)) else [lab]
            f.write("video_%04d %s\n" % (i, " ".join(str(l) for l in lab)))
    maxlab = 1
    with tfrecord.TFRecordWriter(paths_file + ".tfrecord") as w:
        for vid, lab, c in zip(videos, labels, cpv):
            lab = lab if isinstance(lab, (list, tuple)) else [lab]
            maxlab = max(maxlab, len(lab))
            if len(vid) < c * fpc:
                error("video has %d frames, needs %d" % (len(vid), c * fpc))
            for k in range(c * fpc):
                w.write(tfrecord.frame_example(vid[k], lab))
    tfrecord.write_size_file(paths_file + ".tfrecord.size", len(videos), defs.input_mode.video, list(cpv), fpc, maxlab)
    info("Serialized %d videos to %s.tfrecord" % (len(videos), paths_file))


def write_vector_dataset(paths_file, sequences, labels, fpc, clips_per_item=1):
    """Vector-mode dataset (serialize.py:258-266,605-606; `.size` type `vectors`): sequences = list of float32 arrays
    [clips * fpc, dim] (e.g. the word embeddings of a caption), labels = per item either ONE label list (stored with every record,
    like a video's label) or a list of per-record label lists (word-level targets)."""
    cpv = clips_per_item if isinstance(clips_per_item, (list, tuple)) else [clips_per_item] * len(sequences)
    maxlab = 1
    with open(paths_file, "w") as f:
        for i in range(len(sequences)):
            f.write("item_%04d\n" % i)
    with tfrecord.TFRecordWriter(paths_file + ".tfrecord") as w:
        for seq, lab, c in zip(sequences, labels, cpv):
            seq = np.asarray(seq, np.float32)
            if len(seq) != c * fpc:
                error("vector item has %d records, needs %d" % (len(seq), c * fpc))
            per_record = len(lab) == len(seq) and all(isinstance(x, (list, tuple, np.ndarray)) for x in lab)
            for k in range(len(seq)):
                l = list(lab[k]) if per_record else (list(lab) if isinstance(lab, (list, tuple, np.ndarray)) else [lab])
                maxlab = max(maxlab, len(l))
                w.write(tfrecord.vector_example(seq[k], l))
    tfrecord.write_size_file(paths_file + ".tfrecord.size", len(sequences), defs.input_mode.vectors, list(cpv), fpc, maxlab)
    info("Serialized %d vector items to %s.tfrecord" % (len(sequences), paths_file))
