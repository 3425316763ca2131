#!/usr/bin/env python3
"""Times the host side of one 64-clip training batch on this box: TFRecord read + CRC (native reader), pageable vs pinned
host->device copy, next to the device step.  usage: host_pipeline_probe.py <folder made by examples/make_synthetic_dataset.py>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vltf_amd import _hostio


def main():
    path = os.path.join(sys.argv[1], "train.txt.tfrecord")
    shape, n = (240, 320, 3), 1024
    for verify in (True, False):
        off, ts = 0, []
        for _ in range(3):
            t0 = time.perf_counter()
            imgs, labels, off = _hostio.read_frames(path, off, n, shape, verify_crc=verify)
            ts.append(time.perf_counter() - t0)
        print("read %d frames (%.0f MB), crc %s: %s ms" % (n, imgs.nbytes / 1e6, verify, [round(t * 1e3, 1) for t in ts]))
    pinned = torch.empty(imgs.shape, dtype=torch.uint8, pin_memory=True)
    t0 = time.perf_counter()
    _hostio.read_frames(path, 0, n, shape, out=pinned.numpy())
    print("read into pinned memory: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    dev = torch.empty(imgs.shape, dtype=torch.uint8, device="cuda:0")
    for name, src in (("pageable", torch.from_numpy(imgs)), ("pinned", pinned)):
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            dev.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("H2D %s: %s ms (%.1f GB/s)" % (name, [round(t * 1e3, 1) for t in ts], imgs.nbytes / min(ts) / 1e9))


if __name__ == "__main__":
    main()
