#!/usr/bin/env python3
"""Launcher of tests/verify_checkpoint.py (the checkpoint verifier uses the float64 oracle, which is test infrastructure
and therefore lives under tests/): python tools/verify_checkpoint.py --ckpt ... --config ... [--data ...]"""
import os
import runpy
import sys

sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "verify_checkpoint.py")
runpy.run_path(sys.argv[0], run_name="__main__")
