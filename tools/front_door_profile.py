#!/usr/bin/env python3
"""Where FrontDoor.run spends host time (perf_counter around its pieces; same setup as tools/front_door_bench.py)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import ModelCTC, named_config, synth
from efficientconformer_amd import batching
from efficientconformer_amd.batching import FrontDoor

cfg = named_config("EfficientConformerCTCSmall")
m = ModelCTC.from_config(cfg)
sd = synth.make_state_dict(m.encoder.plan, 0, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda(); m.encoder.ragged = True; m.encoder.sub_batches = 3
lens = synth.libri_lengths(1024, seed=4242)
rng = np.random.default_rng(7)
waves = [torch.from_numpy((0.1 * rng.standard_normal(int(n))).astype(np.float32)) for n in lens]
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
door = FrontDoor(device_fn=lambda x, n, hl: m.encode_greedy(x, n, x_len_host=hl)[2:], device="cuda", max_batch=256, workers=8, zero_pad=False)
door.run(waves); torch.cuda.synchronize()
wrap(door, "_stage", "stage (pack + H2D issue)")
wrap(door, "device_fn", "device_fn launch")
wrap(batching, "bucket_batches", "bucket_batches")
orig_pin = torch.Tensor.pin_memory
def pin(self, *a, **k):
    t0 = time.perf_counter(); r = orig_pin(self, *a, **k); T["pin_memory()"] = T.get("pin_memory()", 0.0) + time.perf_counter() - t0; return r
torch.Tensor.pin_memory = pin
t0 = time.perf_counter(); door.run(waves); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("total %.1f ms" % (tot * 1e3))
for k, v in T.items(): print("  %-28s %.1f ms" % (k, v * 1e3))
for k, v in door.stats.items(): print("  stats.%-22s %.1f ms" % (k, v * 1e3))
