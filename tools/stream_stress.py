#!/usr/bin/env python3
"""Stress that made the rare failure of one bit-identity test reproducible and led to the weight-ring wait fix (profiles/r6_108_ring_wait_fix.txt) (tests/test_gpu_encoder.py::test_sub_batch_streams_are_bit_identical_at_librispeech_batch_sizes):
EfficientConformerCTCSmall, B = 65 with one very short utterance, 2 / 3 row ranges against one stream, N iterations in ONE process, optionally with unrelated torch work on
another stream in flight.  Prints where the first differences sit (utterance, rows, size).   python tools/stream_stress.py [iterations] [noise 0/1]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from efficientconformer_amd import synth

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
noise = len(sys.argv) > 2 and sys.argv[2] == "1"
opts = [a.split("=") for a in sys.argv[3:]]                    # library options name=value (bisecting which kernel is the victim)
cfg, model, sd = bench.build_model("EfficientConformerCTCSmall")
model = model.cuda()
for k, v in opts:
    model.encoder.set_option(k, int(v))
B = 65
lens = synth.libri_lengths(B, seed=100 + B)[:B]
lens[-1] = 2000
audio = torch.from_numpy(synth.make_audio(lens, seed=B)).cuda()
ln = torch.from_numpy(lens).cuda()
model.encoder.sub_batches = 1
ref, rl, _ = model.encoder(audio, ln)
again, _, _ = model.encoder(audio, ln)
print("one stream, cold vs warm equal:", torch.equal(ref, again))
side = torch.cuda.Stream()
a = torch.randn(2048, 2048, device="cuda")
bad = 0
for it in range(n_it):
    ns = 2 + it % 2
    model.encoder.sub_batches = ns
    if noise:
        with torch.cuda.stream(side):
            for _ in range(4):
                a = torch.tanh(a @ a * 1e-3)
    got, gl, _ = model.encoder(audio, ln)
    torch.cuda.synchronize()
    if not torch.equal(got, ref):
        bad += 1
        if bad > 3:
            continue
        d = got != ref
        utt = d.flatten(1).any(1).nonzero().flatten().tolist()
        print("iteration %d nsub %d: %d elements differ, utterances %s (lens %s), rows of the first %s, max |d| %.3e" % (
            it, ns, int(d.sum()), utt[:10], [int(lens[b]) for b in utt[:10]], d[utt[0]].any(1).nonzero().flatten().tolist()[:8], float((got - ref).abs().max())))
print("iterations %d, mismatches %d, noise %s, options %s" % (n_it, bad, noise, opts))
