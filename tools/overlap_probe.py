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
    ap.add_argument("--mode", default="sync", choices=["sync", "pipelined"],
                    help="sync: the range's stream waits for its collective, then runs the head (dist.ShardedEncoder default); pipelined: the head of "
                         "a chunk runs one step later, right behind that step's encoder kernels of the same range (ShardedEncoder(pipelined=True))")
    ap.add_argument("--standin", default="spin", choices=["spin", "copies"],
                    help="what stands in for the all-gather on the comm stream: spin = one own-slot copy + an idle wave for bytes received / --xgmi-gbs "
                         "(the DURATION of the transfer; round 3's `copies` = world - 1 device-to-device copies run at 2 TB/s, 7x faster than xGMI)")
    ap.add_argument("--xgmi-gbs", type=float, default=300.0, help="sustained all-gather receive bandwidth per GPU to emulate (GB/s)")
    ap.add_argument("--range-frames", default="", help="share of the valid frames per row range in percent, e.g. 40,35,25 (default: equal)")
    args = ap.parse_args()
    import ctypes as C
    import numpy as np
    from efficientconformer_amd import _lib
    lib = _lib.load_debug()
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
    if args.range_frames:
        fr = np.cumsum(lens_np // 160 + 1).astype(np.float64)
        shares = [float(v) for v in args.range_frames.split(",")]
        acc, cc = 0.0, []
        for v in shares[:-1]:
            acc += v / sum(shares)
            cc.append((int(np.searchsorted(fr, acc * fr[-1])) + 4) // 8 * 8)
        enc.sub_batch_bounds = cc
    pending = {}
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
            key = (lo, hi)
            cur = torch.cuda.current_stream(dev)
            if args.mode == "pipelined" and key in pending:      # the previous step's chunk of this range: consumed here, behind this step's encoder
                pg, plen, pev, prec = pending.pop(key)
                cur.wait_event(pev)
                for r in range(args.world):                       # the head runs on the GATHERED chunk: world x the range's rows
                    model._head(pg[r].float() if pg.dtype != torch.float32 else pg[r], plen)
                he = torch.cuda.Event(enable_timing=True); he.record(cur)
                prec["head"] = he
            issue = torch.cuda.Event(); issue.record(cur)       # this step's collective is issued here (pipelined: behind the previous chunk's head)
            comm.wait_event(issue)
            with torch.cuda.stream(comm):
                cs = torch.cuda.Event(enable_timing=True); cs.record(comm)
                src = out[lo:hi] if wire == torch.float32 else out[lo:hi].to(wire)
                out.record_stream(comm)
                if key not in bufs:
                    bufs[key] = torch.empty((args.world,) + tuple(src.shape), dtype=wire, device=dev)
                g = bufs[key]
                g[0].copy_(src)                                   # this rank's own slot
                if args.standin == "copies":
                    for r in range(1, args.world):                # the bytes the other world - 1 ranks send, as device-to-device copies
                        g[r].copy_(src, non_blocking=True)
                else:                                             # the TIME those bytes take over xGMI: an idle wave on the comm stream
                    us = (args.world - 1) * src.numel() * src.element_size() / (args.xgmi_gbs * 1e9) * 1e6
                    _lib.check(lib.effconf_debug_spin(C.c_double(us), comm.cuda_stream), "spin")
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(comm)
            entry = {"rows": (lo, hi), "done": done, "cs": cs, "ce": ev, "head": ev}
            rec["ranges"].append(entry)
            if args.mode == "pipelined":
                pending[key] = (g, out_len[lo:hi].clone(), ev, entry)
            else:
                # as torch.distributed does for a collective issued on this stream: the range's stream resumes when the chunk has arrived,
                # and the consumer (CTC head on the gathered chunk: world x the range's rows) runs on it
                cur.wait_event(ev)
                for r in range(args.world):
                    model._head(g[r].float() if g.dtype != torch.float32 else g[r], out_len[lo:hi])
                he = torch.cuda.Event(enable_timing=True); he.record(cur)
                entry["head"] = he
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
    print("# mode %s, stand-in %s%s, row ranges %s" % (args.mode, args.standin, "" if args.standin == "copies" else " (%.0f GB/s per GPU)" % args.xgmi_gbs,
                                                       [tuple(r["rows"]) for r in log[2]["ranges"]]))
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
    # share of the collectives' time that lies under encoder kernels: the union over steps and row ranges of [step start, that range's last
    # encoder kernel done] on one time axis (the first probed step's start)
    base = log[2]["t0"]
    busy = []
    for r in log[2:]:
        a = base.elapsed_time(r["t0"])
        for x in r["ranges"]:
            busy.append((a, base.elapsed_time(x["done"])))
    busy.sort()
    merged = []
    for a, b in busy:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    tot = under = 0.0
    for r in log[3:-1]:                 # interior steps: a neighbour on both sides
        for x in r["ranges"]:
            a, b = base.elapsed_time(x["cs"]), base.elapsed_time(x["ce"])
            tot += b - a
            under += sum(max(0.0, min(b, hi) - max(a, lo)) for lo, hi in merged)
    print("# collective time per step %.3f ms; share of it that runs while encoder kernels (any row range, this or the next step) are executing: %.0f %%"
          % (tot / max(len(log) - 4, 1), 100 * under / max(tot, 1e-9)))


if __name__ == "__main__":
    main()
