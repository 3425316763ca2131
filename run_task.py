#!/usr/bin/env python3
"""python3 run_task.py <config.yml> [--gpus N] -- same entry point as the reference (run_task.py:155-160)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vltf_amd.run_task import cli  # noqa: E402

if __name__ == "__main__":
    cli()
