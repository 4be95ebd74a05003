#!/usr/bin/env python3
"""scripts/strong_scaling_probe.py against an experimental build:
    python scripts/exp_strong.py scripts/_build/librt_NAME.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracing_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import strong_scaling_probe  # noqa: E402,F401
