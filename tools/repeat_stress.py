#!/usr/bin/env python3
"""The bench's own configuration (EfficientConformerCTCSmall, B = 256 LibriSpeech-shaped, ragged, 3 row ranges on 3 streams) run N times: every forward must reproduce the
first one bit for bit.   python tools/repeat_stress.py [iterations] [model=NAME] [precision=bf16|split] [batch=B] [name=value library options ...]"""
import sys

import torch

sys.path.insert(0, ".")
import bench
from efficientconformer_amd import synth

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
name, precision, B = "EfficientConformerCTCSmall", "bf16", 256
opts = []
for a in sys.argv[2:]:
    k, v = a.split("=")
    if k == "model": name = v
    elif k == "precision": precision = v
    elif k == "batch": B = int(v)
    else: opts.append((k, v))
cfg, model, sd = bench.build_model(name)
model = model.cuda()
model.encoder.precision = precision
for k, v in opts:
    model.encoder.set_option(k, int(v))
lens = synth.libri_lengths(B, seed=1234)[:B]
lens = -((-lens)).astype(lens.dtype)
order = sorted(range(B), key=lambda i: -int(lens[i]))
lens = lens[order]
audio = torch.from_numpy(synth.make_audio(lens, seed=7)).cuda()
ln = torch.from_numpy(lens).cuda()
enc = model.encoder
enc.ragged, enc.sub_batches = True, 3
ref, rl, _ = enc(audio, ln, x_len_host=lens)
bad = 0
for it in range(n_it):
    got, gl, _ = enc(audio, ln, x_len_host=lens)
    torch.cuda.synchronize()
    if not torch.equal(got, ref):
        bad += 1
        if bad <= 3:
            d = got != ref
            utt = d.flatten(1).any(1).nonzero().flatten().tolist()
            print("iteration %d: %d elements differ, utterances %s, max |d| %.3e" % (it, int(d.sum()), utt[:8], float((got - ref).abs().max())))
print("%s %s B=%d: iterations %d, mismatches %d, options %s" % (name, precision, B, n_it, bad, opts))
