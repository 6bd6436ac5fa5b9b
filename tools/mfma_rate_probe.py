#!/usr/bin/env python3
"""Matrix-pipe rate of v_mfma_f32_32x32x16_{bf16,f16} on this chip: pure MFMA loops (csrc/debug.hip kinds 11 - 14), 1024 workgroups of 4 waves.
Is the f16 form as fast as the bf16 form?  (The split-precision mode, csrc/split.hip, multiplies on the f16 form.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientconformer_amd import _lib  # noqa: E402

lib = _lib.load_debug()
buf = torch.zeros(1 << 16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for kind, name in ((11, "32x32x16 bf16, 4 accumulators"), (12, "32x32x16 f16,  4 accumulators"), (13, "32x32x16 bf16, 1 accumulator (dependent chain)"),
                   (14, "32x32x16 f16,  1 accumulator (dependent chain)")):
    blocks, iters = 2048, 2000
    for _ in range(2):
        _lib.check(lib.effconf_debug_neighbour(kind, blocks, 0, iters, buf.data_ptr(), buf.numel(), st), "probe")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.effconf_debug_neighbour(kind, blocks, 0, iters, buf.data_ptr(), buf.numel(), st), "probe")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flop = blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16
    print("%-50s %8.3f ms  %8.1f TFLOP/s" % (name, ms, flop / ms / 1e9))
