#!/usr/bin/env python3
"""One-GPU stand-in for the multi-GPU step (north_star: "RCCL all-gather of encoder outputs before CTC, overlapped with the last encoder
stage on a side HIP stream"): the default bench step (EfficientConformerCTCSmall, B = 256, three ragged row ranges on three streams)
with `dist.ShardedEncoder`'s exact stream protocol - a `range_hook` that records an event on the range's stream, a comm stream that waits
for it - where the collective itself is replaced by device-to-device copies of the SAME BYTES an 8-rank all-gather makes this rank
receive (7 x the range's rows, fp32 wire by default), followed by the CTC head on the "gathered" chunk on a head stream.
Run under `rocprofv3 --kernel-trace`; tools/overlap_timeline.py turns the trace into profiles/r3_overlap_timeline.txt.

    python tools/overlap_probe.py [--steps 6] [--world 8] [--wire fp32|bf16]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientconformer_amd import ModelCTC, named_config, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--model", default="EfficientConformerCTCSmall")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = named_config(args.model)
    model = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(model.encoder.plan, 0, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    enc = model.encoder
    lens_np = synth.libri_lengths(256, seed=1234)
    audio = torch.from_numpy(synth.make_audio(lens_np, seed=1234)).to(dev)
    lens = torch.from_numpy(lens_np).to(dev)
    cuts = [0, 80, 160, 256]
    pads = [int(lens_np[cuts[i]:cuts[i + 1]].max()) for i in range(3)]
    enc.sub_batches, enc.sub_batch_streams, enc.ragged, enc.stagger_ranges = 3, 3, True, True        # bench.py's default step
    comm, head = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    wire = torch.float32 if args.wire == "fp32" else torch.bfloat16
    bufs = {}

    def step():
        chunks = []

        def hook(lo, hi, out, out_len):
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))          # the range's last encoder kernel is enqueued
            comm.wait_event(done)
            with torch.cuda.stream(comm):
                src = out[lo:hi] if wire == torch.float32 else out[lo:hi].to(wire)
                out.record_stream(comm)
                key = (lo, hi)
                if key not in bufs:
                    bufs[key] = torch.empty((args.world,) + tuple(src.shape), dtype=wire, device=dev)
                g = bufs[key]
                g[0].copy_(src)                                   # this rank's own slot
                for r in range(1, args.world):                    # the bytes the other world - 1 ranks send
                    g[r].copy_(src, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(comm)
            chunks.append((lo, hi, g, out_len, ev))
        enc(audio, lens, range_hook=hook, x_len_host=lens_np)
        head.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(head):
            for lo, hi, g, out_len, ev in chunks:
                head.wait_event(ev)
                model._head(g[0].float() if g.dtype != torch.float32 else g[0], out_len[lo:hi])
        return chunks

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for _ in range(args.steps):
        step()
    torch.cuda.current_stream(dev).wait_stream(head)
    torch.cuda.synchronize()
    nbytes = sum(b.numel() * b.element_size() for b in bufs.values())
    print("overlap_probe: %d steps, stand-in collective bytes per step %.1f MB (world %d, wire %s)" % (args.steps, nbytes / 1e6, args.world, args.wire))


if __name__ == "__main__":
    main()
