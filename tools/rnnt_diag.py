#!/usr/bin/env python3
"""Time the RNN-T greedy decode paths (per-utterance kernel vs cluster decode) on synthetic encoder outputs."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import Transducer, named_config, synth

name = "EfficientConformerTransducerMedium"
cfg = named_config(name)
m = Transducer.from_config(cfg)
sd = synth.make_state_dict(m.encoder.plan, 0, None, prefix="encoder.")
bb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2
sd.update(synth.make_transducer_state_dict(m.encoder.plan.dim_out, cfg["decoder_params"], cfg["joint_params"], 0, blank_bias=bb))
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda()
B, T = 128, 200
g = torch.Generator().manual_seed(0)
f = torch.randn(B, T, 360, generator=g).cuda()
lens = torch.tensor(sorted([int(x) for x in np.linspace(60, T, B)], reverse=True)).cuda()
for mode in (0, 1):
    m.set_decode_option("cluster_decode", mode)
    t, n = m.decode_encoded(f, lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        t, n = m.decode_encoded(f, lens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("mode %d: %.2f ms, tokens %d (max %d), frames %d" % (mode, dt * 1e3, int(n.sum()), int(n.max()), int(lens.sum())))
    if mode == 0: ref = (t.clone(), n.clone())
print("identical:", torch.equal(ref[0], t) and torch.equal(ref[1], n))
