#!/usr/bin/env python3
"""What is mel_kernel sensitive to?  (GPU box; HISTORY.md section 5, profiles/r2_mel_repro.txt)

    python tools/mel_repro.py [trials]

Victim: diagnostic variants of mel_kernel (csrc/mel.hip: 1 canaries around the exchange buffers, 2 self-verifying hand-offs,
4 workgroup barriers instead of wave-level hand-offs; `+96K` = 96 KB of unused dynamic LDS per workgroup) on stream s1.
Aggressor on stream s0: the real subsampling kernels (forward_mel: sublinear_kernel first) or ONE synthetic kernel that loads a
single CU resource (csrc/debug.hip: LDS 16-byte hammer, VALU + transcendental, global loads, global stores, MFMA, LDS publish +
barrier loop, LDS 4-byte hammer).  Every line: aggressor, victim variant, differing mel elements per trial, the victim's counters
(canary words overwritten | own writes lost at hand-off 1/2/3 | exchanged values that changed between two reads at 1/2/3/4)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import bench
from efficientconformer_amd import synth, _lib

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _lib.load_debug()
cfg, model, sd = bench.build_model("EfficientConformerCTCSmall")
enc = model.cuda().encoder
enc.sub_batches = 1
B = 129
lens = synth.libri_lengths(B, seed=100 + B)[:B]
lens[-1] = 2000
audio = torch.from_numpy(synth.make_audio(lens, seed=B)).cuda()
ln = torch.from_numpy(lens).cuda()
a0, l0, a1, l1 = audio[:64].contiguous(), ln[:64].contiguous(), audio[64:].contiguous(), ln[64:].contiguous()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
enc._ensure_packed()
h = enc._handle
tm = a1.shape[1] // 160 + 1
nbuf = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")       # 256 MB playground of the synthetic neighbours


def mel_variant(variant, extra, stream):
    out = torch.empty(a1.shape[0], 80, tm, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
    _lib.check(lib.effconf_debug_mel(h, variant, extra, a1.data_ptr(), a1.shape[0], a1.shape[1], out.data_ptr(), cnt.data_ptr(),
                                     stream.cuda_stream), "debug_mel")
    return out, cnt


def neighbour(kind, blocks, lds, iters, stream):
    _lib.check(lib.effconf_debug_neighbour(kind, blocks, lds, iters, nbuf.data_ptr(), nbuf.numel(), stream.cuda_stream), "neighbour")


ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, stream, reps=5):
    with torch.cuda.stream(stream):
        fn(); a, b = ev(), ev(); a.record()
        for _ in range(reps):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


with torch.cuda.stream(s1):
    want, _ = mel_variant(0, 0, s1)
with torch.cuda.stream(s0):
    mel0, mlen0 = enc.mel_frontend(a0, l0)
    enc.forward_mel(mel0, mlen0)
torch.cuda.synchronize()
prod = enc.mel_frontend(a1, l1)[0]
torch.cuda.synchronize()
print("product mel_kernel == variant 0:", bool(torch.equal(prod, want)))

VARIANTS = [(0, 0, "V0"), (2, 0, "V2 verify"), (3, 0, "V3 canary+verify"), (4, 0, "V4 barriers"), (0, 96 << 10, "V0 +96K"), (0, 24 << 10, "V0 +24K")]
print("---- victim alone (ms per launch, equal to V0, counters)")
for v, extra, name in VARIANTS:
    ms = timed(lambda: mel_variant(v, extra, s1), s1)
    with torch.cuda.stream(s1):
        out, cnt = mel_variant(v, extra, s1)
    torch.cuda.synchronize()
    print("%-18s %.3f ms  equal %s  counters %s" % (name, ms, bool(torch.equal(out, want)), cnt.tolist()))

e0, e1 = ev(), ev()
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)

# synthetic neighbours: (label, kind, blocks, lds bytes); iters calibrated to ~1.5 ms
SYN = [("lds16 80K", 0, 2048, 80000), ("lds16 10K", 0, 4096, 10240), ("valu+trans", 1, 4096, 0), ("gload", 2, 4096, 0), ("gstore", 3, 4096, 0),
       ("mfma", 4, 4096, 0), ("lds16+barrier 80K", 5, 2048, 80000), ("lds16+barrier 10K", 5, 4096, 10240), ("lds4 10K", 6, 4096, 10240)]
cal = {}
for label, kind, blocks, ldsb in SYN:
    t1 = timed(lambda: neighbour(kind, blocks, ldsb, 4, s0), s0, reps=2)
    cal[label] = max(1, int(4 * 1.5 / max(t1, 1e-3)))
    print("neighbour %-20s iters %5d  (%.3f ms at 4 iters)" % (label, cal[label], t1))


def run(aggr_label, aggr_fn, delays_ms):
    for v, extra, name in VARIANTS:
        diffs, cnts = [], torch.zeros(8, dtype=torch.int64)
        for tr in range(trials):
            for d in delays_ms:
                g = ev(); g.record(); s0.wait_event(g); s1.wait_event(g)
                with torch.cuda.stream(s0):
                    torch.cuda._sleep(int(0.05 * cyc_per_ms))
                    aggr_fn()
                with torch.cuda.stream(s1):
                    torch.cuda._sleep(int((0.05 + d) * cyc_per_ms) + 1)
                    out, cnt = mel_variant(v, extra, s1)
                torch.cuda.synchronize()
                diffs.append(int((out != want).sum()))
                cnts += cnt.cpu().long()
        print("%-22s | %-18s | diff/trial %s | canary %d | lost writes %d %d %d | unstable reads %d %d %d %d" % (
            aggr_label, name, diffs, cnts[0], cnts[1], cnts[3], cnts[5], cnts[2], cnts[4], cnts[6], cnts[7]))
    sys.stdout.flush()


print("---- real aggressor: forward_mel on s0 (sublinear_kernel first), victim delayed by 0 .. 0.35 ms")
run("forward_mel", lambda: enc.forward_mel(mel0, mlen0), [0.0, 0.07, 0.14, 0.21, 0.28, 0.35])
print("---- synthetic aggressors (~1.5 ms each), victim delayed by 0.1 / 0.5 ms")
for label, kind, blocks, ldsb in SYN:
    run(label, lambda: neighbour(kind, blocks, ldsb, cal[label], s0), [0.1, 0.5])
