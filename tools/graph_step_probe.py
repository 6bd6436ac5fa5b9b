#!/usr/bin/env python3
"""Is the default bench step bound by host launches?  The same ragged three-range step (encoder + per-range CTC heads) eager vs
captured once into a hipGraph and replayed (no host launches at all).

    python tools/graph_step_probe.py [steps=40]
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import ModelCTC, named_config, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = named_config("EfficientConformerCTCSmall")
m = ModelCTC.from_config(cfg)
sd = synth.make_state_dict(m.encoder.plan, 0, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda(); m.encoder.ragged = True; m.encoder.sub_batches = 3
lens_np = np.sort(synth.libri_lengths(256, seed=1234))[::-1].copy()
audio = torch.from_numpy(synth.make_audio(lens_np, seed=1234)).cuda()
lens = torch.from_numpy(lens_np).cuda()
frames = int((lens_np // 160 + 1).sum())

def run_eager(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = m.encode_greedy(audio, lens, x_len_host=lens_np)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out
run_eager(5)
dt, ref = run_eager(steps)
print("eager: %.3f ms per step -> %.1f M frames/s" % (dt * 1e3, frames / dt / 1e6))
t0 = time.perf_counter()
for _ in range(steps): m.encode_greedy(audio, lens, x_len_host=lens_np)
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
print("host time to enqueue one step: %.3f ms" % (host * 1e3))

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m.encode_greedy(audio, lens, x_len_host=lens_np)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    g_out = m.encode_greedy(audio, lens, x_len_host=lens_np)
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(a, b) for a, b in zip(g_out, ref))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): g.replay()
torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / steps
print("hipGraph replay: %.3f ms per step -> %.1f M frames/s; outputs identical to eager: %s" % (dg * 1e3, frames / dg / 1e6, ok))
