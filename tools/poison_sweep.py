#!/usr/bin/env python3
"""Out-of-bounds / uninitialised-read / repeatability sweep of the HIP forward over every named configuration (GPU box only).

Three checks per configuration, at two utterance lengths, for every subsampler variant:
  guards      EFFCONF_POISON_GUARDS=<KiB>: every packed parameter buffer sits between NaN-filled guard regions (include/effconf.h);
              the output must stay finite and bit-identical to the run without guards;
  workspace   EFFCONF_POISON_WORKSPACE=<byte>: a fresh workspace is filled with 0x00 / 0x7F (huge finite) / 0xFF (NaN) bytes;
              the three outputs must be bit-identical (no kernel may read workspace that nobody wrote);
  repeat      the same forward <reps> times on one instance, every third one next to a GEMM on a second stream: bit-identical.

Round 2 found the strided blocks' conv_res GEMM (180 -> 256, EfficientConformer Medium) reading its last weight row 128 bytes past
the buffer this way; tests/test_gpu_exact_and_sweep.py::test_no_kernel_reads_past_a_parameter_buffer keeps the first check in the
GPU suite.

    python tools/poison_sweep.py [--reps 50] [--guard-kib 4096] [model names ...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from efficientconformer_amd import ModelCTC, config, named_config, synth  # noqa: E402


def build(name, guards="0", fill=""):
    os.environ["EFFCONF_POISON_GUARDS"] = guards
    os.environ["EFFCONF_POISON_WORKSPACE"] = fill
    m = ModelCTC(named_config(name)["encoder_params"], {"vocab_size": 256})
    sd = synth.make_state_dict(m.encoder.plan, 3, 256, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.cuda()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--guard-kib", type=int, default=4096)
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    names = a.names or sorted(config._NAMED)
    side = torch.cuda.Stream()
    w = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
    failures = 0
    for name in names:
        for tm, lens in ((301, [301, 190]), (1001, [1001, 640, 333])):
            mel, ln = synth.make_mel(len(lens), 80, tm, lens, seed=5)
            mel_d, ln_d = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
            ref = {}
            problems = []
            for tag, guards, fill in (("plain", "0", ""), ("guards", str(a.guard_kib), ""), ("ws00", "0", "0"), ("ws7f", "0", "127"),
                                      ("wsff", "0", "255")):
                m = build(name, guards, fill)
                for fs in (0, 1, 2):
                    m.encoder.set_option("fuse_subsample", fs)
                    out = m.encoder.forward_mel(mel_d, ln_d)[0]
                    if tag == "plain":
                        ref[fs] = out.clone()
                        for r in range(a.reps if fs == 2 else 0):
                            if r % 3 == 1:
                                with torch.cuda.stream(side):
                                    for _ in range(1 + r % 5):
                                        w @ w
                            if not torch.equal(m.encoder.forward_mel(mel_d, ln_d)[0], ref[fs]):
                                problems.append("repeat %d differs" % r)
                                break
                    elif not torch.isfinite(out).all():
                        problems.append("%s fs%d: %d non-finite" % (tag, fs, int((~torch.isfinite(out)).sum())))
                    elif not torch.equal(out, ref[fs]):
                        problems.append("%s fs%d: max diff %.3g" % (tag, fs, float((out - ref[fs]).abs().max())))
                del m
                torch.cuda.synchronize()
            failures += bool(problems)
            print("%-36s Tm %4d  %s" % (name, tm, "; ".join(problems) if problems else "clean"), flush=True)
    print("poison sweep: %d failing case(s)" % failures)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
