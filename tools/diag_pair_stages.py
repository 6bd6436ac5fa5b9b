#!/usr/bin/env python3
"""Round 5 diagnostics: per-stage trace of the forward with the column-pair chains (chain2.hip) against chain.hip's - first differing stage, and where
in the row / column space the differences sit.  usage: diag_pair_stages.py <config> <mel frames> <chain_pair> <chain_full_max>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientconformer_amd import ModelCTC, named_config, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "EfficientConformerCTCSmall"
    tm = int(sys.argv[2]) if len(sys.argv) > 2 else 700
    pair = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    full = int(sys.argv[4]) if len(sys.argv) > 4 else 192
    extra = [a.split("=") for a in sys.argv[5:]]
    cfg = named_config(name)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 7, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    lens = [tm, int(tm * 0.77), int(tm * 0.52)]
    mel, ln = synth.make_mel(3, 80, tm, lens, seed=4321 + tm)
    mel, ln = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
    m.encoder.set_option("chain_small_m", 0)
    m.encoder.set_option("chain_full_max", full)
    m.encoder.set_option("chain_pair", 0)
    out0, _, t0 = m.encoder.trace_forward_mel(mel, ln)
    m.encoder.set_option("chain_pair", pair)
    for k, v in extra:
        m.encoder.set_option(k, int(v))
    out1, _, t1 = m.encoder.trace_forward_mel(mel, ln)
    bad = 0
    for k, v in t0.items():
        w = t1.get(k)
        if w is None or tuple(w.shape) != tuple(v.shape):
            print("%-22s missing / shape" % k)
            continue
        a, b = v.float(), w.float()
        neq = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
        n = int(neq.sum())
        if n == 0:
            continue
        bad += 1
        rows = neq.any(dim=1).nonzero().flatten()
        cols = neq.any(dim=0).nonzero().flatten()
        d = (a - b).abs()
        print("%-22s shape %-14s differing %8d  max %.3e nan %d | rows %d..%d (%d) mod128 first %s | cols %d..%d (%d)" % (
            k, tuple(v.shape), n, float(d[~torch.isnan(d)].max()) if n else 0.0, int(torch.isnan(b).sum()), int(rows[0]), int(rows[-1]), rows.numel(),
            sorted(set((rows % 128).tolist()))[:12], int(cols[0]), int(cols[-1]), cols.numel()))
        if bad >= 6:
            break
    print("pair %d full %d: %s (final equal: %s)" % (pair, full, "IDENTICAL" if bad == 0 else "%d stages differ" % bad, bool(torch.equal(out0, out1))))


if __name__ == "__main__":
    main()
