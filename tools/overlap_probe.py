#!/usr/bin/env python3
"""One-GPU stand-in for the multi-GPU step (north_star: "RCCL all-gather of encoder outputs before CTC, overlapped with the last encoder
stage on a side HIP stream"): the default bench step (EfficientConformerCTCSmall, B = 256, three ragged row ranges on three streams)
with `dist.ShardedEncoder`'s exact stream protocol - a `range_hook` that records an event on the range's stream, a comm stream that waits
for it - where the collective itself is replaced by device-to-device copies of the SAME BYTES an 8-rank all-gather makes this rank
receive (7 x the range's rows, fp32 wire by default), followed by the CTC head on the "gathered" chunk on a head stream.
Prints an event timeline (HIP events on the range / comm / head streams, no profiler: under `rocprofv3 --kernel-trace` every launch costs the
host ~40 us, the three ranges' enqueues serialise and the trace shows a 9 ms step that does not exist otherwise - tools/overlap_timeline.py
reads such a trace all the same):  per step, when each range's last encoder kernel finished, when its stand-in collective ran on the comm
stream, when the head consumed it - relative to the step's start - and the same for the NEXT step's first kernels, which the later
collectives run under.

    python tools/overlap_probe.py [--steps 20] [--world 8] [--wire fp32|bf16]
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--model", default="EfficientConformerCTCSmall")
    ap.add_argument("--no-collective", action="store_true", help="bench.py's single-GPU step: no comm stream, the CTC head per range on the range's stream")
    ap.add_argument("--stagger", type=int, default=0, help="1: range 0's stream gets the higher priority (ConformerEncoder.stagger_ranges)")
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
    enc.sub_batches, enc.sub_batch_streams, enc.ragged, enc.stagger_ranges = 3, 3, True, bool(args.stagger)        # bench.py's default step
    comm = torch.cuda.Stream(device=dev)          # stands in for the process group's own stream
    wire = torch.float32 if args.wire == "fp32" else torch.bfloat16
    bufs = {}

    log = []

    def step():
        chunks = []
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record(torch.cuda.current_stream(dev))
        rec = {"t0": t0, "ranges": []}

        def hook(lo, hi, out, out_len):
            done = torch.cuda.Event(enable_timing=True)
            done.record(torch.cuda.current_stream(dev))          # the range's last encoder kernel is enqueued
            if args.no_collective:
                model._head(out[lo:hi], out_len[lo:hi])
                ev = torch.cuda.Event(enable_timing=True); ev.record(torch.cuda.current_stream(dev))
                rec["ranges"].append({"rows": (lo, hi), "done": done, "cs": done, "ce": ev, "head": ev})
                return
            comm.wait_event(done)
            with torch.cuda.stream(comm):
                cs = torch.cuda.Event(enable_timing=True); cs.record(comm)
                src = out[lo:hi] if wire == torch.float32 else out[lo:hi].to(wire)
                out.record_stream(comm)
                key = (lo, hi)
                if key not in bufs:
                    bufs[key] = torch.empty((args.world,) + tuple(src.shape), dtype=wire, device=dev)
                g = bufs[key]
                g[0].copy_(src)                                   # this rank's own slot
                for r in range(1, args.world):                    # the bytes the other world - 1 ranks send
                    g[r].copy_(src, non_blocking=True)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(comm)
            # as torch.distributed does for a collective issued on this stream: the range's stream resumes when the chunk has arrived,
            # and the consumer (CTC head on the gathered chunk) runs on it
            torch.cuda.current_stream(dev).wait_event(ev)
            model._head(g[0].float() if g.dtype != torch.float32 else g[0], out_len[lo:hi])
            he = torch.cuda.Event(enable_timing=True); he.record(torch.cuda.current_stream(dev))
            rec["ranges"].append({"rows": (lo, hi), "done": done, "cs": cs, "ce": ev, "head": he})
            chunks.append((lo, hi, g, out_len, ev))
        enc(audio, lens, range_hook=hook, x_len_host=lens_np)
        te = torch.cuda.Event(enable_timing=True); te.record(torch.cuda.current_stream(dev))      # the forward's join on the caller's stream
        rec["head_end"], rec["join"] = te, te
        log.append(rec)
        return chunks

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    nbytes = sum(b.numel() * b.element_size() for b in bufs.values())
    if args.no_collective:
        bufs.update({r["rows"]: torch.empty(0) for r in log[0]["ranges"]})
    print("# overlap_probe: %d steps, stand-in collective bytes per step %.1f MB (world %d, wire %s); times in ms from the step's start (median over steps)"
          % (args.steps, nbytes / 1e6, args.world, args.wire))
    import statistics as st
    steps = log[2:]
    med = lambda f: st.median(f(r) for r in steps)
    nr = len(steps[0]["ranges"])
    print("# %-22s %10s %12s %12s %12s %12s" % ("row range", "encoder", "collective", "collective", "head done", "bytes"))
    print("# %-22s %10s %12s %12s %12s %12s" % ("", "done", "start", "end", "(its stream)", "MB"))
    for i in range(nr):
        lo, hi = steps[0]["ranges"][i]["rows"]
        print("  rows [%3d, %3d)%8s %10.3f %12.3f %12.3f %12.3f %12.1f" % (lo, hi, "", med(lambda r: r["t0"].elapsed_time(r["ranges"][i]["done"])),
              med(lambda r: r["t0"].elapsed_time(r["ranges"][i]["cs"])), med(lambda r: r["t0"].elapsed_time(r["ranges"][i]["ce"])),
              med(lambda r: r["t0"].elapsed_time(r["ranges"][i]["head"])),
              bufs[(lo, hi)].numel() * bufs[(lo, hi)].element_size() * (args.world - 1) / args.world / 1e6))
    print("  join on the caller's stream      %10.3f" % med(lambda r: r["t0"].elapsed_time(r["join"])))
    nxt = [log[k]["t0"].elapsed_time(log[k + 1]["t0"]) for k in range(2, len(log) - 1)]
    print("  next step starts at              %10.3f   (= step period)" % st.median(nxt))
    cov = []
    for k in range(2, len(log) - 1):
        r, n = log[k], log[k + 1]
        last_done = max(r["t0"].elapsed_time(x["done"]) for x in r["ranges"])
        period = r["t0"].elapsed_time(n["t0"])
        tot = under = 0.0
        for x in r["ranges"]:
            a, b = r["t0"].elapsed_time(x["cs"]), r["t0"].elapsed_time(x["ce"])
            tot += b - a
            # under this step's encoder kernels: before the last range's last encoder kernel finished
            under += max(0.0, min(b, last_done) - a)
        cov.append(under / max(tot, 1e-9))
    print("# share of the collectives' time that runs while encoder kernels of another row range of the same step are executing: %.0f %%" % (100 * st.median(cov)))


if __name__ == "__main__":
    main()
