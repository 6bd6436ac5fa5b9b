#!/usr/bin/env python3
"""End-to-end rate of the batching front door: host waveforms (pageable memory, as a data loader hands them over) -> greedy label ids,
PCIe inclusive, against the device-resident rate of the same batches.

    python tools/front_door_bench.py [utterances=1024] [max_batch=256]

EfficientConformerCTCSmall, LibriSpeech-shaped lengths, ragged batches on three streams (bench.py's default configuration)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import ModelCTC, named_config, synth  # noqa: E402
from efficientconformer_amd.batching import FrontDoor, bucket_batches  # noqa: E402

n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = named_config("EfficientConformerCTCSmall")
m = ModelCTC.from_config(cfg)
sd = synth.make_state_dict(m.encoder.plan, 0, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda()
m.encoder.ragged = True
m.encoder.sub_batches = 3
lens = synth.libri_lengths(n_utt, seed=4242)
rng = np.random.default_rng(7)
waves = [torch.from_numpy((0.1 * rng.standard_normal(int(n))).astype(np.float32)) for n in lens]
frames = int(sum(int(n) // 160 + 1 for n in lens))
mb = sum(int(n) for n in lens) * 4 / 1e6
print("utterances %d, %.1f h of audio, %.0f MB of fp32 samples, %d mel frames" % (n_utt, sum(lens) / 16000 / 3600, mb, frames))

# ---- device-resident reference: the same batches, inputs already in HBM
plan = bucket_batches([int(w.numel()) for w in waves], max_batch)
dev = []
for idx in plan:
    n = [int(waves[i].numel()) for i in idx]
    x = torch.zeros(len(idx), max(n))
    for r, i in enumerate(idx):
        x[r, :n[r]] = waves[i]
    dev.append((x.cuda(), torch.tensor(n).cuda()))
hls = [l.cpu().numpy() for _, l in dev]
for (x, l), hl in list(zip(dev, hls))[:1]:
    m.encode_greedy(x, l, x_len_host=hl)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    for (x, l), hl in zip(dev, hls):
        m.encode_greedy(x, l, x_len_host=hl)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("device-resident (encoder + CTC head, %d batches): %.1f ms -> %.1f M frames/s" % (len(dev), dt * 1e3, frames / dt / 1e6))
del dev

# ---- the front door: pageable host waveforms in, label id lists out
def timed(name, door):
    ref = door.run(waves)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        out = door.run(waves)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        assert out == ref
    print("%s: %.1f ms -> %.1f M frames/s end to end (%.1f GB/s of samples over PCIe)" % (name, best * 1e3, frames / best / 1e6, mb / 1e3 / best))
    return ref


a = timed("front door, fn = model.greedy_labels (one sync per batch), 8 packing threads", FrontDoor(m.greedy_labels, "cuda", max_batch=max_batch, zero_pad=False))
for w in (1, 4, 8, 16):
    b = timed("front door, device_fn = encode_greedy (no sync inside the loop), %d packing threads" % w,
              FrontDoor(device_fn=lambda x, n, hl: m.encode_greedy(x, n, x_len_host=hl)[2:], device="cuda", max_batch=max_batch, workers=w, zero_pad=False))
    assert a == b
