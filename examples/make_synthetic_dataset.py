#!/usr/bin/env python3
"""Writes a small UCF-101-shaped dataset in the reference's on-disk layout (paths file + .tfrecord + .tfrecord.size;
serialize.py:138-151,246-256): random 240x320x3 BGR frames, 16 frames per clip, labels in [0, 101).
usage: make_synthetic_dataset.py <folder> [videos] [clips_per_video] [--captions <steps> <dim> <vocab>]
--captions also writes <name>_words.txt(.tfrecord, .size): a `vectors` dataset with one random word vector + next-word id per record,
paired item by item with the videos (examples/encoder_decoder_description.yml)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vltf_amd import serialize


def main():
    folder = sys.argv[1]
    args = [a for a in sys.argv[2:]]
    captions = None
    if "--captions" in args:
        i = args.index("--captions")
        captions = tuple(int(x) for x in args[i + 1:i + 4])
        args = args[:i]
    nvid = int(args[0]) if len(args) > 0 else 128
    cpv = int(args[1]) if len(args) > 1 else 1
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(0)
    for name, count in (("train", nvid), ("val", max(1, nvid // 4))):
        videos = [rng.integers(0, 256, (16 * cpv, 240, 320, 3), dtype=np.uint8) for _ in range(count)]
        labels = [int(l) for l in rng.integers(0, 101, count)]
        serialize.write_video_dataset(os.path.join(folder, name + ".txt"), videos, labels, fpc=16, clips_per_video=cpv)
        if captions:
            steps, dim, vocab = captions
            seqs = [rng.standard_normal((cpv * steps, dim)).astype(np.float32) for _ in range(count)]
            targets = [[[int(t)] for t in rng.integers(0, vocab, cpv * steps)] for _ in range(count)]
            serialize.write_vector_dataset(os.path.join(folder, name + "_words.txt"), seqs, targets, steps, cpv)
    print("wrote %s/{train,val}.txt(.tfrecord, .tfrecord.size)" % folder)


if __name__ == "__main__":
    main()
