#!/usr/bin/env python3
"""Timing ablations of the fused split FFN kernel (csrc/sxf_ffn.hip) through libeffconf_debug.so: what each phase of a chunk costs.
    python tools/sxf_ffn_probe.py [--model EfficientConformerCTCSmall] [--rows 48512]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import ModelCTC, _lib, named_config, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="EfficientConformerCTCSmall")
    ap.add_argument("--rows", type=int, default=0)
    args = ap.parse_args()
    dlib = _lib.load_debug()
    cfg = named_config(args.model)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 3, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.encoder.precision = "split"
    m = m.cuda()
    # a second handle inside the diagnostic library (the product library's handle is opaque to it)
    enc = m.encoder
    real = _lib.load
    _lib.load = lambda: dlib
    try:
        enc._packed = False
        enc._handle = None
        enc._ensure_packed()
    finally:
        _lib.load = real
    h = enc._handle
    plan = enc.plan
    seen = set()
    print("%-8s %6s %8s | %s" % ("block", "D", "rows", "us per launch: full | -G1 | -swish | -G2 | -stream | -barrier | -G1-G2 | -all"))
    for k, b in enumerate(plan.blocks):
        for which, D in ((1, b.dim_model), (2, b.dim_expand)):
            if D in seen:
                continue
            seen.add(D)
            rows = args.rows or {120: 48512, 168: 24320, 240: 12160}.get(D, 16384)
            x = torch.randn(rows, D, device="cuda")
            y = torch.empty_like(x)
            res = []
            for abl in (0, 1, 2, 4, 8, 16, 5, 31):
                for _ in range(3):
                    _lib.check(dlib.effconf_debug_sxf_ffn(h, k, which, x.data_ptr(), rows, y.data_ptr(), 0, abl, None), "ffn", dlib)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    _lib.check(dlib.effconf_debug_sxf_ffn(h, k, which, x.data_ptr(), rows, y.data_ptr(), 0, abl, None), "ffn", dlib)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 20 * 1000)
            print("%-8s %6d %8d | %s" % ("%d.%d" % (k, which), D, rows, " | ".join("%7.1f" % r for r in res)))
    enc._handle = None      # the diagnostic library's handle is not destroyed through the product library


if __name__ == "__main__":
    main()
