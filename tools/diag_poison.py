"""Where does a poisoned (0xFF = NaN patterns) fresh workspace reach a stage output?  EFFCONF_POISON_WORKSPACE=255 python tools/diag_poison.py [config]

Runs the split-precision traced forward (per-module kernels) and the chain forward of a config on a fresh, poisoned workspace and prints, per traced
stage, how many non-finite values it holds and in which rows / columns.  Diagnostic only (no oracle): a kernel that reads workspace nobody wrote shows
up as the first stage with a non-zero count.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import synth                                   # noqa: E402
from efficientconformer_amd.config import named_config                    # noqa: E402
from efficientconformer_amd.model_ctc import ModelCTC                     # noqa: E402


def model(name, seed, precision):
    cfg = dict(named_config(name), model_type="CTC")
    vocab = min(256, cfg["tokenizer_params"]["vocab_size"])
    m = ModelCTC(cfg["encoder_params"], {"vocab_size": vocab})
    sd = synth.make_state_dict(m.encoder.plan, seed, vocab, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.encoder.precision = precision
    return m.cuda()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Tiny"
    tm, lens = 100, [100, 77, 52]
    mel, ln = synth.make_mel(3, 80, tm, lens, seed=4421)
    for precision in ("split", "bf16"):
        m = model(name, 7, precision)
        out, out_len, got = m.encoder.trace_forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda())
        print("== %s %s traced forward: out non-finite %d of %d" % (name, precision, int((~torch.isfinite(out)).sum()), out.numel()))
        first = True
        for k, t in got.items():
            bad = ~torch.isfinite(t)
            if bad.any():
                rows = bad.any(1).nonzero().flatten().tolist()
                cols = bad.any(0).nonzero().flatten().tolist()
                print("   %-28s %6d bad of %s  rows %s cols %s" % (k, int(bad.sum()), tuple(t.shape), rows[:12], cols[:12]))
                if first or k.endswith((".q", ".k", ".v")) and k.startswith("blocks.0"):
                    first = first and k.endswith((".q", ".k", ".v"))
                    for r in rows:
                        print("      row %4d: %s" % (r, "".join("x" if v else "." for v in bad[r].tolist())))
        for chain in (0, 1):
            for ragged in (False, True):
                m = model(name, 7, precision)
                enc = m.encoder
                if precision == "split":
                    enc.set_option("split_chain", chain)
                enc.ragged = ragged
                o, ol, _ = enc.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda(), x_len_host=ln if ragged else None)
                bad = ~torch.isfinite(o)
                where = [(b, bad[b].any(1).nonzero().flatten().tolist()[:8]) for b in range(o.shape[0]) if bad[b].any()]
                print("   forward chain %d ragged %d: non-finite %d  out_len %s  (utterance, rows) %s" % (chain, ragged, int(bad.sum()), ol.tolist(), where))
                if precision != "split":
                    break


if __name__ == "__main__":
    main()
