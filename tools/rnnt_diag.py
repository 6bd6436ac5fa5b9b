#!/usr/bin/env python3
"""Time the RNN-T greedy decode paths (per-utterance kernel vs cluster decode, both workgroup -> XCD mappings) on synthetic encoder outputs.

    [RNNT_MODES=cluster:by_slice:shape,...] python tools/rnnt_diag.py [blank_bias=1.2] [batch=128]      shape: 0 = 8 x 8 x 2, 1 = 16 x 16 x 1
"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientconformer_amd import Transducer, named_config, synth

name = "EfficientConformerTransducerMedium"
cfg = named_config(name)
m = Transducer.from_config(cfg)
sd = synth.make_state_dict(m.encoder.plan, 0, None, prefix="encoder.")
bb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sd.update(synth.make_transducer_state_dict(m.encoder.plan.dim_out, cfg["decoder_params"], cfg["joint_params"], 0, blank_bias=bb))
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda()
T = 200
g = torch.Generator().manual_seed(0)
f = torch.randn(B, T, 360, generator=g).cuda()
lens = torch.tensor(sorted([int(x) for x in np.linspace(60, T, B)], reverse=True)).cuda()
ref = None
modes = [tuple(int(v) for v in mm.split(':')) for mm in os.environ.get('RNNT_MODES', '0:0:0,1:1:0,1:1:1').split(',')]
for mode, by_slice, shape in modes:
    m.set_decode_option("cluster_decode", mode)
    m.set_decode_option("cluster_by_slice", by_slice)
    m.set_decode_option("cluster_shape", shape)
    t, n = m.decode_encoded(f, lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        t, n = m.decode_encoded(f, lens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("B %d blank_bias %.1f cluster %d by_slice %d shape %d: %.2f ms, tokens %d (max %d), frames %d" % (B, bb, mode, by_slice, shape, dt * 1e3, int(n.sum()), int(n.max()), int(lens.sum())))
    if ref is None: ref = (t.clone(), n.clone())
    else: print("   identical to the per-utterance kernel:", torch.equal(ref[0], t) and torch.equal(ref[1], n))
