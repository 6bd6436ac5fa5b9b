#!/usr/bin/env python3
"""Small-batch latency by kernel route (round 4): the row-stationary chains stream ALL weights of a chain through every
workgroup - a fixed ~40 - 90 us of dependent chunk steps per launch however few rows there are.  Per-GEMM / tiled kernels split N across
workgroups instead.  EfficientConformerCTCSmall, 10 s utterances, B = 1 / 4 / 16, encoder forward only:

    python tools/small_batch_routes.py [--only B] [--opts name=value,name=value]     (one route, for profiling)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402
import bench                # noqa: E402
from efficientconformer_amd import synth     # noqa: E402

ROUTES = [("default (chains; launches <= 4096 rows on 2-wave workgroups)", {}), ("chain_small_m=0 (round 3: 8-wave workgroups for every launch)", {"chain_small_m": 0}),
          ("chain_small_m=16384", {"chain_small_m": 16384}), ("fuse_chain=0", {"fuse_chain": 0}), ("chain_max_dim=192 (D = 240 stage per GEMM)", {"chain_max_dim": 192}),
          ("chain_max_dim=128", {"chain_max_dim": 128})]


def measure(model, B, reps=50):
    lens = np.full(B, 160000, dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=1)).cuda()
    ln = torch.from_numpy(lens).cuda()
    for _ in range(5):
        enc, el, _ = model.encoder(audio, ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        enc, el, _ = model.encoder(audio, ln)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, enc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=0)
    ap.add_argument("--opts", default="")
    a = ap.parse_args()
    if a.only:
        _, model, _ = bench.build_model("EfficientConformerCTCSmall")
        model = model.cuda()
        for kv in filter(None, a.opts.split(",")):
            k, v = kv.split("=")
            model.encoder.set_option(k, int(v))
        ms, _ = measure(model, a.only, reps=20)
        print("B=%d %s: %.3f ms" % (a.only, a.opts or "default", ms))
        return
    ref = {}
    for name, opts in ROUTES:
        _, model, _ = bench.build_model("EfficientConformerCTCSmall")
        model = model.cuda()
        for k, v in opts.items():
            model.encoder.set_option(k, v)
        row = []
        for B in (1, 4, 16):
            ms, enc = measure(model, B)
            if B not in ref:
                ref[B] = enc.float().clone()
            err = float((enc.float() - ref[B]).abs().max())
            row.append("B=%d %.3f ms (max |diff| to the default route %.3g)" % (B, ms, err))
        print("%-70s %s" % (name, "   ".join(row)), flush=True)
        del model


if __name__ == "__main__":
    main()
