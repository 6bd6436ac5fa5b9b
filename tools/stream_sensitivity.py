#!/usr/bin/env python3
"""Which kernels of ANOTHER stream disturb the mel frontend?  (GPU box; HISTORY.md section 5, profiles/r1_16_stream_sensitivity.txt)

    python tools/stream_sensitivity.py [default|nochain|nofusesub|frommel|melmel] [victim: mel|blocks]

Stream s1 runs ConformerEncoder.mel_frontend on 65 utterances, delayed by 0 .. 2.9 ms in 0.1 ms steps; stream s0 runs, from a common
time origin, the aggressor:
    default    the full forward from audio (mel, fused conv-subsampling + Linear, 15 blocks)
    nochain    the same with fuse_chain = 0 (unfused block kernels)
    nofusesub  the same with fuse_subsample = 0 (separate conv and GEMM kernels)
    frommel    forward_mel (no mel kernel on s0: the subsampling kernels come first)
    melmel     two mel_frontend launches
With `blocks` as the second argument the victim on s1 is forward_mel (everything AFTER the mel boundary) instead of the mel kernel.
Each line gives both streams' [start, end] in ms and the number of mel elements that differ from the same launch run alone.
Finding (round 1): the mel output differs exactly when mel_kernel overlaps the other stream's SUBSAMPLING kernels (sublinear_kernel,
or subsample_conv_kernel + gemm_kernel), never next to another mel_kernel or next to the block kernels; frame pairs are hit as a
unit, a few FFT bins each.  ConformerEncoder therefore forks its sub-batch streams at the mel boundary."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from efficientconformer_amd import synth, _lib

kind = sys.argv[1] if len(sys.argv) > 1 else "default"
victim = sys.argv[2] if len(sys.argv) > 2 else "mel"
cfg, model, sd = bench.build_model("EfficientConformerCTCSmall")
enc = model.cuda().encoder
enc.sub_batches = 1
if kind == "nochain":
    enc.set_option("fuse_chain", 0)
if kind == "nofusesub":
    enc.set_option("fuse_subsample", 0)
B = 129
lens = synth.libri_lengths(B, seed=100 + B)[:B]
lens[-1] = 2000
audio = torch.from_numpy(synth.make_audio(lens, seed=B)).cuda()
ln = torch.from_numpy(lens).cuda()
a0, l0, a1, l1 = audio[:64].contiguous(), ln[:64].contiguous(), audio[64:].contiguous(), ln[64:].contiguous()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    mel1, mlen1 = enc.mel_frontend(a1, l1)
    want = enc.forward_mel(mel1, mlen1)[0] if victim == "blocks" else mel1
with torch.cuda.stream(s0):
    mel0, mlen0 = enc.mel_frontend(a0, l0)
    enc(a0, l0)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
e0, e1 = ev(), ev()
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
for d in range(30):
    g, a, b, c, e = ev(), ev(), ev(), ev(), ev()
    g.record(); s0.wait_event(g); s1.wait_event(g)
    with torch.cuda.stream(s1):
        torch.cuda._sleep(int(0.1 * d * cyc_per_ms) + 1)
        c.record(); m = enc.forward_mel(mel1, mlen1)[0] if victim == "blocks" else enc.mel_frontend(a1, l1)[0]; e.record()
    with torch.cuda.stream(s0):
        torch.cuda._sleep(int(0.05 * cyc_per_ms))
        a.record()
        if kind == "frommel":
            enc.forward_mel(mel0, mlen0)
        elif kind == "melmel":
            enc.mel_frontend(a0, l0); enc.mel_frontend(a0, l0)
        else:
            enc(a0, l0)
        b.record()
    torch.cuda.synchronize()
    diff = m != want
    print("%-9s s0=[%.2f,%.2f] %s on s1=[%.2f,%.2f] differing elements %d in %d utterances" % (
        kind, g.elapsed_time(a), g.elapsed_time(b), "forward_mel" if victim == "blocks" else "mel", g.elapsed_time(c), g.elapsed_time(e), int(diff.sum()), int(diff.flatten(1).any(1).sum())))
