#!/usr/bin/env python3
"""Stage 2 of tools/mel_repro.py (GPU box): WHICH instruction class of the victim is disturbed by WHICH aggressor.

Victims: self-contained kernels running chains of one instruction class (csrc/debug.hip victim_kernel), the product mel_kernel
(built without packed-fp32 VALU instructions, as all of libeffconf) and its build WITH packed fp32 (round 1's kernel).  Aggressors (another stream): MFMA-only kernels (bf16 32x32x16, fp32 32x32x2, bf16
16x16x32, 50 % duty), a packed-fp32 VALU hammer, at full and at one-workgroup-per-CU occupancy.
Every line: aggressor | victim | differing output elements per trial (run alone = reference)."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from efficientconformer_amd import synth, _lib

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _lib.load_debug()
cfg, model, sd = bench.build_model("EfficientConformerCTCSmall")
enc = model.cuda().encoder
enc._ensure_packed()
h = enc._handle
B = 65
lens = synth.libri_lengths(129, seed=229)[64:129]
audio = torch.from_numpy(synth.make_audio(lens, seed=129)).cuda()
tm = audio.shape[1] // 160 + 1
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
nbuf = torch.zeros(1 << 20, dtype=torch.float32, device="cuda")
ev = lambda: torch.cuda.Event(enable_timing=True)


def mel_variant(variant, extra, stream):
    out = torch.empty(B, 80, tm, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
    _lib.check(lib.effconf_debug_mel(h, variant, extra, audio.data_ptr(), B, audio.shape[1], out.data_ptr(), cnt.data_ptr(), stream.cuda_stream), "debug_mel")
    return out


VB = 2048          # victim workgroups


def victim(kind, iters, stream):
    out = torch.empty(VB * 256 * 16, dtype=torch.float32, device="cuda")
    _lib.check(lib.effconf_debug_victim(kind, VB, iters, out.data_ptr(), stream.cuda_stream), "victim")
    return out


def neighbour(kind, blocks, iters, stream):
    _lib.check(lib.effconf_debug_neighbour(kind, blocks, 0, iters, nbuf.data_ptr(), nbuf.numel(), stream.cuda_stream), "neighbour")


def timed(fn, stream, reps=3):
    with torch.cuda.stream(stream):
        fn(); a, b = ev(), ev(); a.record()
        for _ in range(reps):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


e0, e1 = ev(), ev()
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)

AGGR = [("mfma bf16 32x32x16, 4096 wg", 4, 4096), ("mfma bf16 32x32x16, 256 wg", 4, 256), ("mfma bf16 32x32x16, 1024 wg", 4, 1024),
        ("mfma f32 32x32x2, 4096 wg", 7, 4096), ("mfma bf16 16x16x32, 4096 wg", 9, 4096), ("mfma bf16 50% duty, 4096 wg", 10, 4096),
        ("v_pk_fma_f32 hammer, 4096 wg", 8, 4096), ("valu+trans, 4096 wg", 1, 4096)]
if len(sys.argv) > 2:      # quick mode: only the MFMA aggressors that disturbed mel_kernel
    AGGR = [AGGR[0], AGGR[4], AGGR[5]]
cal = {}
for label, kind, blocks in AGGR:
    t = timed(lambda: neighbour(kind, blocks, 8, s0), s0, 2)
    cal[label] = max(1, int(8 * 2.0 / max(t, 1e-3)))
    print("aggressor %-30s iters %6d (%.3f ms at 8)" % (label, cal[label], t))

VICT = [("v_fma_f32", 0), ("v_pk_fma_f32", 1), ("v_pk_mul/add_f32", 2), ("v_log/v_exp_f32", 3), ("v_mul/v_add_f32", 4), ("integer mad", 5), ("LDS wave exchange", 6),
        ("pk_fma op_sel_hi:[1,0,1]", 7), ("pk_add neg_lo/hi:[0,1]", 8), ("pk_add op_sel:[0,1]", 9), ("pk_mul op_sel:[0,1] hi:[1,0]", 10),
        ("pk_fma neg:[0,0,1]", 11), ("pk_mov_b32 op_sel:[1,0]", 12), ("pk_add/mul op_sel+neg mix", 13)]
if len(sys.argv) > 2:      # quick mode: only the MFMA aggressors that disturbed mel_kernel
    pass
vit, vref = {}, {}
for name, k in VICT:
    t = timed(lambda: victim(k, 256, s1), s1, 2)
    vit[name] = max(8, int(256 * 0.4 / max(t, 1e-3)))            # ~0.4 ms
    with torch.cuda.stream(s1):
        a = victim(k, vit[name], s1); b = victim(k, vit[name], s1)
    torch.cuda.synchronize()
    vref[name] = a
    print("victim %-20s iters %6d  alone twice equal: %s  finite: %s" % (name, vit[name], bool(torch.equal(a, b)), bool(torch.isfinite(a).all())))
with torch.cuda.stream(s1):
    mref = {v: mel_variant(v, 0, s1) for v in (0, 8)}
torch.cuda.synchronize()
for v in (0, 8):
    print("mel_kernel V%d alone: %.3f ms per launch (B = %d)" % (v, timed(lambda: mel_variant(v, 0, s1), s1, 5), B))
print("mel V0 (product: no packed fp32) vs V8 (packed fp32) alone: %d differing elements (different instruction selection, not an error)" % int((mref[0] != mref[8]).sum()))


def overlap(aggr, vfn):
    label, kind, blocks = aggr
    g = ev(); g.record(); s0.wait_event(g); s1.wait_event(g)
    with torch.cuda.stream(s0):
        torch.cuda._sleep(int(0.05 * cyc_per_ms))
        neighbour(kind, blocks, cal[label], s0)
    with torch.cuda.stream(s1):
        torch.cuda._sleep(int(0.3 * cyc_per_ms))
        out = vfn()
    torch.cuda.synchronize()
    return out


for aggr in AGGR:
    for name, k in VICT:
        d = [int((overlap(aggr, lambda: victim(k, vit[name], s1)) != vref[name]).sum()) for _ in range(trials)]
        print("%-30s | %-20s | diff/trial %s" % (aggr[0], name, d))
    for v in (0, 8):
        outs = [overlap(aggr, lambda: mel_variant(v, 0, s1)) for _ in range(trials)]
        d = [int((o != mref[v]).sum()) for o in outs]
        mx = max(float((o - mref[v]).abs().max()) for o in outs)
        print("%-30s | %-20s | diff/trial %s  max |diff| %.3g" % (aggr[0], ("mel_kernel product" if v == 0 else "mel_kernel +packed fp32"), d, mx))
    sys.stdout.flush()
