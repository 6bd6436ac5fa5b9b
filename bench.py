#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s through the EfficientConformerCTC-Small encoder (+ CTC greedy head).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (ConformerEncoder.forward: mel frontend -> conv subsampling -> 15
Conformer blocks, then fc + argmax + CTC collapse) over one synthetic LibriSpeech-shaped batch that is
already resident in HBM (default: 256 utterances per GPU, `--streams 2`).  `--streams 2` runs the batch as two contiguous row
ranges on two HIP streams inside ConformerEncoder.forward (`sub_batches`, efficientconformer_amd/encoders.py): every kernel of the
path is a one-round launch that alternates HBM-bound load/store bursts with compute, and a second stream fills one's bursts with the
other's compute (+10-13 %, bit-identical to `--streams 1`; the mel frontend stays one launch: DESIGN.md section 5).  One process
per GPU; utterances shard across ranks with no data-path collective inside the timed loop except the all-gather of encoder outputs
(RCCL) on a side stream, overlapped with the CTC head, as north_star asks.
Rank 0 prints ONE JSON line.  `value` counts VALID (un-padded) mel frames of all ranks per second.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from efficientconformer_amd import ModelCTC, Transducer, _lib, named_config, synth  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PROF_CLASSES = ["mel", "subsample_conv", "gemm_ffn", "gemm_other", "layernorm", "attention", "dwconv", "misc"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="EfficientConformerCTCSmall")
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--workload", default="libri", choices=["libri", "fixed"],
                    help="libri: lognormal LibriSpeech-shaped lengths (SURVEY.md 8d W-libri); fixed: 10 s each")
    ap.add_argument("--streams", type=int, default=2,
                    help="ConformerEncoder.sub_batches: contiguous row ranges of the batch on concurrent HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


RNNT_BLANK_BIAS = 1.2         # synthetic joint bias of the blank: ~1 token per 3 encoder frames (synth.make_transducer_state_dict)


def build_model(name):
    cfg = named_config(name)
    if cfg["model_type"] == "Transducer":
        model = Transducer.from_config(cfg)
        sd = synth.make_state_dict(model.encoder.plan, 0, None, prefix="encoder.")
        sd.update(synth.make_transducer_state_dict(model.encoder.plan.dim_out, cfg["decoder_params"], cfg["joint_params"], 0,
                                                   blank_bias=RNNT_BLANK_BIAS))
    else:
        model = ModelCTC.from_config(cfg)
        sd = synth.make_state_dict(model.encoder.plan, 0, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return cfg, model, sd


def make_batch(args, rank, world=1):
    if args.workload == "fixed":
        lens = np.full(args.batch, 160000, dtype=np.int64)
        lmax = 160000
    else:
        lens = synth.libri_lengths(args.batch, seed=1234 + rank)
        # every rank pads to the longest utterance of the GLOBAL batch (seeds are known, so no exchange is needed): the
        # all-gather of encoder outputs needs one shape on all ranks, exactly as a data-parallel collate would give it
        lmax = max(int(synth.libri_lengths(args.batch, seed=1234 + r).max()) for r in range(world))
    audio = synth.make_audio(lens, seed=1234 + rank)
    if audio.shape[1] < lmax:
        audio = np.pad(audio, ((0, 0), (0, lmax - audio.shape[1])))
    return audio, lens


def head(model, enc, enc_len):
    if isinstance(model, Transducer):        # RNN-T greedy (transducer.py:139-186); <= 128 utterances per call: the cluster decode's automatic range
        out = None
        for i in range(0, enc.shape[0], 128):
            out = model.decode_encoded(enc[i:i + 128].contiguous(), enc_len[i:i + 128].contiguous())
        return out
    _, labels, label_len = model._head(enc, enc_len)                 # fc + argmax + CTC collapse (model_ctc.py:90-133)
    return labels, label_len


def step(model, audio, lens):
    enc, enc_len, _ = model.encoder(audio, lens)
    labels, label_len = head(model, enc, enc_len)
    return enc, enc_len, labels, label_len


def cpu_baseline(sd, plan, audio_np, lens_np, budget_s=12.0):
    """The oracle (CPU port of the reference path, oracle/ref_encoder.py) on the host cores, bounded sample."""
    from oracle import ref_encoder as R
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: torch.from_numpy(v) for k, v in sd.items()}
    idx = np.linspace(0, len(lens_np) - 1, 4).round().astype(int)      # 4 utterances spread over the sorted batch
    lens = torch.from_numpy(lens_np[idx].copy())
    audio = torch.from_numpy(audio_np[idx][:, :int(lens.max())].copy())
    frames = int((lens // plan.hop_length + 1).sum())
    threads = torch.get_num_threads()

    rnnt = "decoder.embedding.weight" in sd
    if rnnt:
        from oracle import ref_transducer as RT

    def run():
        with torch.no_grad():
            x, l = R.encoder(audio, lens, osd, plan)
            if rnnt:
                return RT.greedy_decode(osd, x, l, 5)
            return R.ctc_greedy(R.ctc_logits(x, osd), l)
    run()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s or n < 2:
        run()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": frames * n / dt, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": "4 of the %d utterances of rank 0's batch (evenly spaced over the length-sorted batch), %d forward passes "
                      "of the fp32 torch oracle in %.1f s" % (len(lens_np), n, dt)}


def pmc_traffic(args, kernel_prefix):
    """HBM bytes per launch of the dominant kernel class from the committed PMC passes (profiles/pmc_traffic.json, written
    by tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  Counters
    cannot be read from inside the timed process, so this is only filled when the run matches the profiled configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    if (d.get("model"), d.get("batch"), d.get("workload"), d.get("streams", 1)) != (args.model, args.batch, args.workload, max(args.streams, 1)):
        return None
    import re
    calls = tot = 0
    for name, k in d["kernels"].items():
        if re.search(kernel_prefix, name):
            calls += k["calls"]
            tot += k["calls"] * (k["fetch_x2_bytes"] + k["write_bytes"])
    return tot / calls if calls else None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (HIP path only)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg, model, sd = build_model(args.model)
    model = model.to(dev)
    plan = model.encoder.plan
    audio_np, lens_np = make_batch(args, rank, world)
    audio, lens = torch.from_numpy(audio_np).to(dev), torch.from_numpy(lens_np).to(dev)
    valid_frames = int((lens_np // plan.hop_length + 1).sum())
    padded_frames = int(args.batch * (audio_np.shape[1] // plan.hop_length + 1))

    # Sub-batch streams live in the library's host layer (ConformerEncoder.sub_batches): the batch runs as `--streams` contiguous
    # row ranges on concurrent HIP streams and is joined before forward() returns.
    model.encoder.sub_batches = max(args.streams, 1)
    gather_buf = None
    side = torch.cuda.Stream(device=dev) if world > 1 else None

    def gather(enc, producer):
        # all-gather of the encoder outputs over RCCL/xGMI on a side stream, overlapped with the head that follows
        nonlocal gather_buf
        if gather_buf is None:
            gather_buf = torch.empty((world,) + tuple(enc.shape), dtype=torch.bfloat16, device=dev)
        side.wait_stream(producer)
        with torch.cuda.stream(side):
            e16 = enc.to(torch.bfloat16)
            enc.record_stream(side)
            dist.all_gather_into_tensor(gather_buf, e16)

    def full_step():
        cur = torch.cuda.current_stream(dev)
        if world > 1:
            # the previous step's all-gather (normally long finished under the head) is complete before this step's mel kernel starts:
            # mel launches are kept away from kernels of other streams they were not swept against
            cur.wait_stream(side)
        enc, enc_len, _ = model.encoder(audio, lens)
        if world > 1:
            gather(enc, cur)
        labels, _ = head(model, enc, enc_len)
        return labels

    for _ in range(args.warmup):
        full_step()
    if world > 1:
        torch.cuda.current_stream(dev).wait_stream(side)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    if world > 1:
        torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([elapsed, float(valid_frames), float(padded_frames)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
    all_valid, all_padded = float(tot[1]), float(tot[2])

    result = None
    if rank == 0:
        result = {
            "metric": "audio-frames/sec through encoder, EffConformerCTC-Small, 1/2/4/8 GPU",
            "value": all_valid * args.steps / elapsed, "unit": "mel-frames/s (valid, 10 ms hop)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s: %s, B=%d utterances/GPU, %s lengths, audio in HBM -> encoder out + greedy %s"
                                   % (args.model, "bf16 operands / fp32 accumulate", args.batch,
                                      "lognormal 1.5-16 s (LibriSpeech-shaped), sorted desc, zero-padded to the batch max"
                                      if args.workload == "libri" else "10 s",
                                      "RNN-T token ids (synthetic blank bias %.1f)" % RNNT_BLANK_BIAS if isinstance(model, Transducer) else "CTC labels"),
                       "global_batch": args.batch * world, "streams_per_gpu": args.streams, "padded_frames_per_s": all_padded * args.steps / elapsed,
                       "parallelism": "dp%d (utterance shards, all-gather of encoder outputs)" % world},
        }

    # ---- roofline leg: the same steps again with every launch bracketed by HIP events on the launch stream
    if rank == 0 and not args.no_roofline:
        lib = _lib.load()
        h = model.encoder._handle
        _lib.check(lib.effconf_profile_enable(h, 1), "profile_enable")
        nprof = min(args.steps, 5)
        nsub = max(args.streams, 1)

        def read_classes():
            torch.cuda.synchronize()
            out = {}
            for ci, cname in enumerate(PROF_CLASSES):
                ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
                _lib.check(lib.effconf_profile_read(h, ci, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "profile_read")
                out[cname] = {"ms_per_step": ms.value / nprof, "launches_per_step": n.value / nprof,
                              "gflop_per_step": fl.value / nprof / 1e9, "alg_mb_per_step": by.value / nprof / 1e6}
            return out
        # Launch durations are taken with the step's row ranges one after the other on ONE stream: same launch shapes as the timed
        # region, no neighbour kernel on the chip.  (With S streams in flight an event pair on one stream also brackets the time
        # its kernel queues behind the other stream's: that figure is reported as `in_flight`, it is not a kernel duration.)
        model.encoder.sub_batches = 1
        for _ in range(nprof):
            for i in range(nsub):
                lo, hi = args.batch * i // nsub, args.batch * (i + 1) // nsub
                step(model, audio[lo:hi], lens[lo:hi])
        per = read_classes()
        model.encoder.sub_batches = nsub
        flight = None
        if nsub > 1:
            _lib.check(lib.effconf_profile_enable(h, 1), "profile_enable")
            for _ in range(nprof):
                step(model, audio, lens)
            flight = read_classes()["gemm_ffn"]
        _lib.check(lib.effconf_profile_enable(h, 0), "profile_enable")
        dom = per["gemm_ffn"]
        n_l = max(dom["launches_per_step"], 1)
        avg_ms = dom["ms_per_step"] / n_l
        ach = dom["gflop_per_step"] / max(dom["ms_per_step"], 1e-9)          # GFLOP/ms == TFLOP/s
        if isinstance(model, Transducer):      # decode leg on its own (torch events: it is launched on torch's current stream)
            enc, enc_len, _ = model.encoder(audio, lens)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nprof):
                toks, tok_len = model.decode_encoded(enc, enc_len)
            e1.record()
            torch.cuda.synchronize()
            per["rnnt_greedy"] = {"ms_per_step": e0.elapsed_time(e1) / nprof, "launches_per_step": 2,
                                  "tokens_per_step": int(tok_len.sum()), "encoder_frames_per_step": int(enc_len.sum())}
        result["roofline"] = {"kernel": "chain_kernel A (pointwise-2 + FFN2 + block norm + next FFN1 + attention pre-norm + QKV in one pass over the rows; "
                                        "ffn_fused_kernel / gemm_kernel where a chain is not supported)",
                              "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(args, r"chain_kernel<\d+, \d+, \d+, [123],|ffn_fused_kernel"),
                              "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)",
                              "alg_bytes_per_launch": 1e6 * dom["alg_mb_per_step"] / n_l,
                              "avg_launch_ms": avg_ms, "launches_per_step": n_l,
                              "alg_gflop_per_launch": dom["gflop_per_step"] / n_l,
                              "alg_hbm_gbs": dom["alg_mb_per_step"] / max(dom["ms_per_step"], 1e-9),
                              "note": "HIP events around every launch of the class on the launch stream, %d extra steps after the timed region "
                                      "(the step's %d row ranges one after the other on one stream, so launches do not overlap)" % (nprof, nsub)}
        if flight:
            result["roofline"]["in_flight"] = {"event_bracket_ms": flight["ms_per_step"] / max(flight["launches_per_step"], 1),
                                               "note": "event pairs around the same launches with the %d row ranges in flight on %d streams, as in the timed "
                                                       "region: includes the time a kernel queues behind the other stream's kernel" % (nsub, nsub)}
        result["kernel_classes"] = per

    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (task contract)
        result["cpu_baseline"] = cpu_baseline(sd, plan, audio_np, lens_np)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
