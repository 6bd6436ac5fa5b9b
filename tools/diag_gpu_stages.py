#!/usr/bin/env python3
"""First-contact diagnostics on the GPU box: per-stage error table of the HIP path vs the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo root
sys.path.insert(0, ROOT)
from efficientconformer_amd import ModelCTC, named_config, synth  # noqa: E402
from oracle import ref_encoder as R  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Tiny"
    tm = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    cfg = named_config(name)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 7, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    if len(sys.argv) > 3:
        m.encoder.set_option("fuse_chain", int(sys.argv[3]))
    if len(sys.argv) > 4:
        m.encoder.set_option("fuse_subsample", int(sys.argv[4]))
    if len(sys.argv) > 5:
        m.encoder.precision = sys.argv[5]
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}
    lens = [tm, int(tm * 0.77), int(tm * 0.52)]
    mel, ln = synth.make_mel(3, 80, tm, lens, seed=4321 + tm)
    trace = {}
    with torch.no_grad():
        ref, ref_len = R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(ln), osd, m.encoder.plan, trace)
    out, out_len, got = m.encoder.trace_forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda())
    print("out_len", out_len.cpu().tolist(), ref_len.tolist())
    refs = {"subsample": trace["subsample"].transpose(1, 2).reshape(-1, trace["subsample"].shape[1]),
            "linear": trace["linear"].reshape(-1, trace["linear"].shape[-1])}
    for k, v in trace.items():
        if k.startswith("blocks."):
            refs[k] = v.reshape(-1, v.shape[-1])
    for k, v in got.items():
        if k in refs and tuple(v.shape) != tuple(refs[k].shape):
            print("%-22s shape %s vs ref %s" % (k, tuple(v.shape), tuple(refs[k].shape)))
        elif k in refs:
            d = (v.double() - refs[k].double()).abs()
            print("%-22s shape %-14s max %.4e mean %.4e  refmax %.3f nan %d" % (
                k, tuple(v.shape), d.max(), d.mean(), refs[k].abs().max(), int(torch.isnan(v).sum())))
        else:
            print("%-22s shape %-14s (no ref) absmax %.3f nan %d" % (k, tuple(v.shape), v.abs().max(), int(torch.isnan(v).sum())))
    d = (out.cpu() - ref).abs()
    print("FINAL max %.4e mean %.4e" % (d.max(), d.mean()))


if __name__ == "__main__":
    main()
