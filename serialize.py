#!/usr/bin/env python3
"""python3 serialize.py <config.yml> -- same entry point as the reference's serializer (serialize.py:885-902): reads the
`serialize:` block, selects clips / frames per video, writes <paths>.tfrecord + .size (+ side files) and validates them."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vltf_amd.serialize import main  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("init_file", help="Configuration .yml file with a `serialize:` block.")
    main(parser.parse_args().init_file)
