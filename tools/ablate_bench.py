#!/usr/bin/env python3
"""Timing-only experiments: run bench.py against an alternative build of the library (EFFCONF_ABLATE_LIB=<path to .so>).

Used for ablation builds of one kernel (results are WRONG by construction: always with --no-check); never part of the product path.
"""
import os
import runpy
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import efficientconformer_amd._lib as L  # noqa: E402

alt = os.environ.get("EFFCONF_ABLATE_LIB")
if alt:
    L.LIB_PATH = os.path.abspath(alt)
sys.argv = [os.path.join(os.path.dirname(here), "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
