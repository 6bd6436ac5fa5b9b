#!/usr/bin/env python3
"""Timing-only library for tools/ablate_bench.py: the product objects with encoder.hip and attention2.hip recompiled under -DEFFCONF_ABLATE (EFFCONF_SKIP =
bit mask of kernel families whose launches are dropped; EFFCONF_ATTN_ABLATE = bit mask of the attention kernel's parts - the product kernel has no such branches).  Written to efficientconformer_amd/build/libeffconf_ablate.so; never loaded by the package."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientconformer_amd import _build as B  # noqa: E402

B.build()
objdir = os.path.join(B.HERE, "build")
REBUILT = ("encoder.hip", "attention2.hip", "chain2.hip", "chain3.hip")      # chain2 / chain3: their in-kernel phase profilers (EFFCONF_CHAIN{2,3}_PHASES) exist in this library only
abl = []
for src in REBUILT:
    obj = os.path.join(objdir, src.replace(".hip", "_ablate.o"))
    subprocess.check_call([B._hipcc()] + B.FLAGS + B.NO_PACKED_FP32 + ["-DEFFCONF_ABLATE", "-DEFFCONF_PHASE_PROF", "-c", os.path.join(B.CSRC, src), "-o", obj])
    abl.append(obj)
objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES if s not in REBUILT] + abl
out = os.path.join(objdir, "libeffconf_ablate.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print("built", out)
