#!/usr/bin/env python3
"""Small-batch latency of ConformerEncoder.forward (Small, 10 s utterances): eager launches vs a captured HIP graph replay.
Measured (round 1): B = 1 / 4 / 16: 2.14 / 2.17 / 2.21 ms eager, 2.30 / 2.33 / 2.38 ms as a graph - the forward is capturable (no
allocation or synchronisation inside) but the floor is the ~130 dependent kernels x ~16 us each on the GPU side, not host launches."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import bench
from efficientconformer_amd import synth
cfg, model, sd = bench.build_model("EfficientConformerCTCSmall")
model = model.cuda()
for B in (1, 4, 16):
    lens = np.full(B, 160000, dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=1)).cuda(); ln = torch.from_numpy(lens).cuda()
    for _ in range(5): enc, el, _ = model.encoder(audio, ln)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): enc, el, _ = model.encoder(audio, ln)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
    # graph
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): model.encoder(audio, ln)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            genc, gel, _ = model.encoder(audio, ln)
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 50
        ok = torch.equal(genc, enc)
    except Exception as e:
        gr, ok = float("nan"), repr(e)[:200]
    print("B=%d eager %.3f ms  graph %.3f ms  equal=%s" % (B, eager * 1e3, gr * 1e3, ok))
