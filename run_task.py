#!/usr/bin/env python3
"""python3 run_task.py <config.yml> -- same entry point as the reference (run_task.py:155-160)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vltf_amd.run_task import main  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("init_file", help="Configuration .yml file for the run.")
    main(parser.parse_args().init_file)
