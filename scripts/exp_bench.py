#!/usr/bin/env python3
"""bench.py against an experimental build of the library:
    python scripts/exp_bench.py scripts/_build/librt_NAME.so [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracing_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
