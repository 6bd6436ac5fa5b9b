#!/usr/bin/env python3
"""How much of the default step is the fork / join bubble at its boundaries?  (round 4)

bench.py's step joins its three row-range streams before it returns, and the next step forks from that join: at every step boundary the chip
drains (the last kernels of the slowest range run alone) and refills (three mel kernels start together).  rocprofv3's kernel trace serialises the
queues (8.6 ms per step under the tracer), so the bubble cannot be read from a trace; this probe measures it by removing it: K steps on the
default workload with ONE batch in flight (bench.py's loop) against TWO (two encoder handles - separate workspaces - whose forwards are enqueued
alternately on two caller streams; each forward still forks and joins its own three streams, but the neighbour step fills its bubbles).

    GPU_MAX_HW_QUEUES=8 python tools/inflight_probe.py [--steps 40] [--ranges 3]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # six range streams: above the runtime's default of four hardware queues
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--ranges", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="EfficientConformerCTCSmall")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    class A:
        workload = "libri"; batch = a.batch
    audio_np, lens_np = bench.make_batch(A, 0, 1)
    audio, lens = torch.from_numpy(audio_np).to(dev), torch.from_numpy(lens_np).to(dev)
    models = []
    for _ in range(3):
        _, m, _ = bench.build_model(a.model)
        m = m.to(dev)
        m.encoder.sub_batches = a.ranges
        m.encoder.sub_batch_streams = a.ranges
        m.encoder.ragged = True
        models.append(m)
    frames = int((lens_np // models[0].encoder.plan.hop_length + 1).sum())
    callers = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = [None] * 3

    def run(n, inflight):
        for i in range(n):
            k = i % inflight
            with torch.cuda.stream(callers[k]):
                outs[k] = models[k].encode_greedy(audio, lens, x_len_host=lens_np)

    ref = None
    for inflight in (1, 2, 3, 1, 2):
        run(a.warmup, inflight)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.steps, inflight)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        labs = [outs[k][2].cpu() for k in range(inflight)]
        if ref is None:
            ref = labs[0]
        same = all(torch.equal(l, ref) for l in labs)
        print("batches in flight %d: %.3f ms per step, %.2f M frames/s, labels equal to the one-in-flight run: %s" % (inflight, dt * 1e3, frames / dt / 1e6, same), flush=True)


if __name__ == "__main__":
    main()
