"""-m gpu tests added in round 3: per-kernel C-ABI entries against the reference's per-module outputs, the benchmarked path at its
own size, workspace / cache hazards named by the round-2 advisor, hipGraph capture of a forward.
Tolerances: bf16-operand kernels 2 % of the tensor's magnitude (max) / 0.3 % (mean), as tests/test_gpu_encoder.py; "bit-identical"
means torch.equal."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from efficientconformer_amd import ModelCTC, _lib, named_config, synth
from oracle import ref_encoder as R

pytestmark = pytest.mark.gpu
# bf16 path with streaming contexts / causal attention: a query sees few keys (left_context 6: seven), so the softmax averages less of the
# operand rounding away - measured worst 0.067 max / 0.0097 mean (causal, left_context 6) against 0.039 / 0.008 without contexts
STREAM_MAX, STREAM_MEAN = 0.09, 0.012


def _model(name, seed, precision="bf16"):
    cfg = named_config(name)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, seed, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.encoder.precision = precision
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}
    return m.cuda(), osd


def _rel(got, ref):
    d = (got.double() - ref.double()).abs()
    scale = max(float(ref.abs().max()), 1.0)
    return float(d.max()) / scale, float(d.mean()) / scale


# ------------------------------------------------------------------ per-kernel C ABI entries vs the reference's module outputs
@pytest.mark.parametrize("gname", ["tiny_T47.npz", "tiny_T100.npz"])
def test_per_kernel_entries_vs_reference_module_outputs(golden_dir, gname):
    """effconf_subsample / effconf_ffn / effconf_conv_module / effconf_layernorm_residual, each alone, fed with the REFERENCE's own
    module inputs (rebuilt from the per-module outputs the golden holds: blocks.py:119-137) and compared with the reference's output
    of that module (hooks on feed_forward_module1/2, convolution_module, the block: tools/make_goldens.py)."""
    g = np.load(os.path.join(golden_dir, gname))
    m, sd = _model("Tiny", int(g["weight_seed"]))
    enc, lib = m.encoder, _lib.load()
    enc._ensure_packed()
    h = enc._handle
    plan = enc.plan
    b = len(g["mel_len"])
    tm = int(g["mel_len"].max())
    mel, _ = synth.make_mel(b, 80, tm, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.effconf_module_workspace_bytes(h, b, tm), dtype=torch.uint8, device="cuda")

    def t(key):
        return torch.from_numpy(g["trace/" + key]).cuda()
    lin = t("linear")                                                        # (B, T1, D0)
    y = torch.empty_like(lin)
    _lib.check(lib.effconf_subsample(h, torch.from_numpy(mel).cuda().data_ptr(), b, tm, y.data_ptr(), ws.data_ptr(), ws.numel(), st), "subsample")
    worst = {"linear": _rel(y.cpu(), lin.cpu())}
    x = lin
    for k, bp in enumerate(plan.blocks):
        p = "blocks.%d" % k
        rows = x.shape[0] * x.shape[1]
        # ---- FFN1: y = x + 1/2 ffn1(x)
        y = torch.empty_like(x)
        _lib.check(lib.effconf_ffn(h, k, 1, x.contiguous().data_ptr(), rows, y.data_ptr(), ws.data_ptr(), ws.numel(), st), "ffn")
        worst[p + ".ffn1"] = _rel((2.0 * (y - x)).cpu(), t(p + ".ffn1").cpu())
        x1 = x + 0.5 * t(p + ".ffn1")
        x2 = x1 + t(p + ".mhsa")
        # ---- conv module (no residual)
        conv_ref = t(p + ".conv")
        yc = torch.empty_like(conv_ref)
        _lib.check(lib.effconf_conv_module(h, k, x2.contiguous().data_ptr(), x2.shape[0], x2.shape[1], yc.data_ptr(), ws.data_ptr(), ws.numel(), st), "conv_module")
        worst[p + ".conv"] = _rel(yc.cpu(), conv_ref.cpu())
        if bp.transition:        # conv_res = Conv1d(D -> De, k = 1, stride 2) on frames 0, 2, 4, ... (blocks.py:106-110)
            w = torch.from_numpy(sd[p + ".conv_res.1.weight"]).cuda()[:, :, 0]
            res = x2[:, ::bp.conv_stride] @ w.t() + torch.from_numpy(sd[p + ".conv_res.1.bias"]).cuda()
        else:
            res = x2
        x3 = res + conv_ref
        # ---- FFN2
        rows3 = x3.shape[0] * x3.shape[1]
        y = torch.empty_like(x3)
        _lib.check(lib.effconf_ffn(h, k, 2, x3.contiguous().data_ptr(), rows3, y.data_ptr(), ws.data_ptr(), ws.numel(), st), "ffn")
        worst[p + ".ffn2"] = _rel((2.0 * (y - x3)).cpu(), t(p + ".ffn2").cpu())
        # ---- block output = LayerNorm(x3 + 1/2 ffn2)
        out_ref = t(p + ".out")
        yo = torch.empty_like(out_ref)
        _lib.check(lib.effconf_layernorm_residual(h, k, 4, x3.contiguous().data_ptr(), t(p + ".ffn2").contiguous().data_ptr(), C.c_float(0.5), rows3,
                                                  yo.data_ptr(), st), "layernorm_residual")
        mx, mean = _rel(yo.cpu(), out_ref.cpu())
        assert mx < 1e-5, (p, mx)                                           # fp32 kernel
        x = out_ref
    torch.cuda.synchronize()
    print({k: ("%.4f" % v[0], "%.5f" % v[1]) for k, v in worst.items()})
    for k, (mx, mean) in worst.items():
        assert mx < 0.02 and mean < 0.003, (k, mx, mean)


# ------------------------------------------------------------------ advisor, round 2: precision modes sharing one workspace
@pytest.mark.parametrize("exact", ["fp32", "split"])
def test_alternating_precision_modes_on_one_shape_keep_the_bf16_path_bit_identical(exact):
    """fp32 -> bf16 -> fp32 -> bf16 on one handle, one shape, one stream (= one workspace): the exact-mode forward lays its buffers
    over the workspace in which the bf16 path cached its positional projections E; the second bf16 run must recompute them and be
    bit-equal to the first (round 2: it read fp32 activations as E)."""
    m, _ = _model("Tiny", 7, exact)
    enc = m.encoder
    mel, ln = synth.make_mel(3, 80, 100, [100, 77, 52], seed=11)
    mel_d, ln_d = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
    outs = []
    for prec in (exact, "bf16", exact, "bf16", "bf16"):
        enc.precision = prec
        out, _, _ = enc.forward_mel(mel_d, ln_d)
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[2])
    assert torch.equal(outs[1], outs[3]) and torch.equal(outs[1], outs[4])
    assert float((outs[0] - outs[1]).abs().max()) < 0.1


# ------------------------------------------------------------------ advisor, round 2: chunk loads behind the last key row
@pytest.mark.parametrize("name,tm", [("EfficientConformerCTCSmall", 383), ("EfficientConformerCTCSmall", 1279), ("EfficientConformerCTCMedium", 383)])
@pytest.mark.parametrize("fill", ["255", "127"])
def test_last_key_row_chunk_loads_do_not_read_the_workspace_slack(name, tm, fill, monkeypatch):
    """B = 1 (the longest utterance is the last one: nothing masks its last key group) with a stage-1 grouped length that is a multiple of
    the 64-key block (Tm = 383 -> T1 = 192 -> Tg = 64; 1279 -> 640 -> 214 is the control) and a head width that is not a multiple of 8
    (d = 90 / 135): the 16-byte chunk that closes the last head's last key row ends in the never-written slack of the K / E buffers.
    A workspace pre-filled with NaN patterns (255) or huge values (127) must give finite output identical to a zero-filled one."""
    mel, ln = synth.make_mel(1, 80, tm, [tm], seed=9)
    mel_d, ln_d = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
    outs = {}
    for f in ("0", fill):
        monkeypatch.setenv("EFFCONF_POISON_WORKSPACE", f)
        m, _ = _model(name, 3)
        out, _, _ = m.encoder.forward_mel(mel_d, ln_d)
        assert torch.isfinite(out).all(), (name, tm, f)
        outs[f] = out.cpu()
        del m
    assert torch.equal(outs["0"], outs[fill])


# ------------------------------------------------------------------ the benchmarked path at the benchmark's size
def test_bench_default_workload_trimmed_ranges_on_three_streams_equal_each_range_alone():
    """bench.py's default step - EfficientConformerCTCSmall, B = 256 LibriSpeech-shaped utterances, three trimmed row ranges on three
    streams, CTC head per range (ModelCTC.encode_greedy) - against every range run ALONE on one stream as its own batch: encoder
    output, lengths and greedy labels bit for bit; and a sample of each range against the oracle (the reference path on that
    collated sub-batch) within the bf16 tolerance (0.06 max / 0.010 mean)."""
    m, sd = _model("EfficientConformerCTCSmall", 0)
    enc = m.encoder
    lens = synth.libri_lengths(256, seed=1234)
    audio = torch.from_numpy(synth.make_audio(lens, seed=1234)).cuda()
    ln = torch.from_numpy(lens).cuda()
    cuts = [0, 80, 160, 256]
    pads = [int(lens[cuts[i]:cuts[i + 1]].max()) for i in range(3)]
    enc.sub_batches, enc.sub_batch_streams, enc.trim_sub_batches = 3, 3, True
    for _ in range(2):                                                      # twice: the second pass runs with warm caches / recycled buffers
        out, out_len, labels, label_len = m.encode_greedy(audio, ln, range_pad=pads)
    torch.cuda.synchronize()
    enc.sub_batches, enc.trim_sub_batches = 1, False
    for i in range(3):
        lo, hi = cuts[i], cuts[i + 1]
        alone, alone_len, _ = enc(audio[lo:hi, :pads[i]].contiguous(), ln[lo:hi].contiguous())
        _, lab, n = m._head(alone, alone_len)
        ti = alone.shape[1]
        assert torch.equal(out[lo:hi, :ti], alone) and torch.equal(out_len[lo:hi], alone_len), i
        assert float(out[lo:hi, ti:].abs().sum()) == 0.0
        assert torch.equal(label_len[lo:hi], n) and torch.equal(labels[lo:hi, :ti], lab), i
        # oracle on 2 utterances of the range collated with the range's LONGEST one (pad frames are live: the padded length matters)
        rows = [lo, lo + (hi - lo) // 2, hi - 1]
        sub = audio[rows, :pads[i]].cpu()
        with torch.no_grad():
            ref, ref_len = R.encoder(sub, ln[rows].cpu(), sd, enc.plan)
        got, got_len, _ = enc(audio[rows, :pads[i]].contiguous(), ln[rows].contiguous())
        d = (got.cpu() - ref).abs()
        assert ref_len.tolist() == got_len.cpu().tolist() and float(d.max()) < 0.06 and float(d.mean()) < 0.010, (i, float(d.max()), float(d.mean()))


# ------------------------------------------------------------------ hipGraph capture of a forward
@pytest.mark.parametrize("name", ["Tiny", "EfficientConformerCTCSmall"])
def test_forward_is_graph_capturable_and_replays_bit_identically(name):
    """include/effconf.h promises that a forward only enqueues kernels (no allocation, synchronisation or host <-> device copy): capture
    effconf_encoder_forward + effconf_ctc_greedy into a hipGraph on a side stream, replay it on new input contents, compare with eager."""
    m, _ = _model(name, 5)
    enc = m.encoder
    lens = np.array([40000, 33000, 21000, 16000], dtype=np.int64)
    a0 = torch.from_numpy(synth.make_audio(lens, seed=1)).cuda()
    a1 = torch.from_numpy(synth.make_audio(lens, seed=2)).cuda()
    ln = torch.from_numpy(lens).cuda()
    eager = []
    for a in (a0, a1):
        out, out_len, _ = enc(a, ln)
        _, lab, n = m._head(out, out_len)
        eager.append((out.clone(), lab.clone(), n.clone()))
    static_in = a0.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        enc(static_in, ln)                                                  # warm-up on the capture stream: workspace allocation happens here
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        g_out, g_len, _ = enc(static_in, ln)
        _, g_lab, g_n = m._head(g_out, g_len)
    for a, (out, lab, n) in ((a1, eager[1]), (a0, eager[0]), (a1, eager[1])):
        static_in.copy_(a)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g_out, out) and torch.equal(g_lab, lab) and torch.equal(g_n, n)
    # an eager forward after the replays (same stream key as the warm-up) still agrees: the capture did not leave a stale cache tag
    with torch.cuda.stream(s):
        out, _, _ = enc(a0, ln)
    torch.cuda.synchronize()
    assert torch.equal(out, eager[0][0])


# ------------------------------------------------------------------ ragged batches
def _ragged_vs_alone(m, sd, audio, lens, nsub, from_mel=False, oracle=True, tol=(0.06, 0.010)):
    enc = m.encoder
    ln = torch.from_numpy(lens).cuda()
    x = audio.cuda()
    enc.ragged, enc.sub_batches, enc.trim_sub_batches = True, nsub, False
    fwd = enc.forward_mel if from_mel else enc
    out, out_len, _ = fwd(x, ln, x_len_host=lens)
    out2, _, _ = fwd(x, ln)                                             # lengths fetched from the device; second pass: warm caches
    assert torch.equal(out, out2)
    enc.ragged, enc.sub_batches = False, 1
    for b in range(len(lens)):
        li = int(lens[b])
        xb = (x[b:b + 1, :, :li] if from_mel else x[b:b + 1, :li]).contiguous()
        alone, alone_len, _ = fwd(xb, ln[b:b + 1].contiguous())
        tb = int(alone_len[0])
        assert int(out_len[b]) == tb and alone.shape[1] == tb
        assert torch.equal(out[b, :tb], alone[0]), (b, float((out[b, :tb] - alone[0]).abs().max()))
        assert float(out[b, tb:].abs().sum()) == 0.0
        if oracle:
            with torch.no_grad():
                ref, ref_len = (R.encoder_from_mel(xb.cpu(), ln[b:b + 1].cpu(), sd, enc.plan) if from_mel else R.encoder(xb.cpu(), ln[b:b + 1].cpu(), sd, enc.plan))
            d = (alone.cpu() - ref).abs()
            assert ref_len.tolist() == [tb] and float(d.max()) < tol[0] and float(d.mean()) < tol[1], (b, float(d.max()), float(d.mean()))
    return out, out_len


@pytest.mark.parametrize("nsub", [1, 2, 3])
def test_ragged_batch_equals_every_utterance_run_alone_tiny(nsub):
    """ConformerEncoder.ragged: every utterance at its own length in one concatenated row space - its output must be, bit for bit, the
    encoder's output for that utterance ALONE (B = 1, the rectangular path) and within the bf16 tolerance of the oracle (the reference
    on that utterance alone); lengths with T % 3 = 0, 1, 2 in the grouped stage, one- and multi-range forwards."""
    m, sd = _model("Tiny", 7)
    lens = np.array([48000, 47840, 41000, 37000, 30160, 22000, 12000, 9000, 3000, 640, 300], dtype=np.int64)     # down to 2 mel frames
    audio = torch.from_numpy(synth.make_audio(lens, seed=4))
    _ragged_vs_alone(m, sd, audio, lens, nsub)


@pytest.mark.parametrize("name", ["EfficientConformerCTCSmall", "EfficientConformerCTCMedium", "EfficientConformerCTCLarge",
                                  "EfficientConformerTransducerSmall", "EfficientConformerTransducerLarge"])
def test_ragged_batch_equals_every_utterance_run_alone_shipped_configs(name):
    m, sd = _model(name, 3)
    lens = np.array([70000, 52345, 33000, 20000, 8000], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=6))
    _ragged_vs_alone(m, sd, audio, lens, 2, oracle=(name == "EfficientConformerCTCSmall"))


def test_ragged_batch_from_mel_and_greedy_labels():
    """The mel entry point (the parity boundary) in ragged mode, and ModelCTC.encode_greedy on ragged ranges = the head on the joined output."""
    m, sd = _model("Tiny", 7)
    lens = np.array([100, 93, 77, 52, 31, 9], dtype=np.int64)
    mel, _ = synth.make_mel(6, 80, 100, lens.tolist(), seed=21)
    _ragged_vs_alone(m, sd, torch.from_numpy(mel), lens, 2, from_mel=True)
    enc = m.encoder
    enc.ragged, enc.sub_batches = True, 2
    ln = torch.from_numpy(lens).cuda()
    out, out_len, labels, label_len = m.encode_greedy(torch.from_numpy(mel).cuda(), ln, from_mel=True, x_len_host=lens)
    _, lab, n = m._head(out, out_len)
    assert torch.equal(labels, lab) and torch.equal(label_len, n)


def test_ragged_bench_size_batch_vs_utterances_alone():
    """B = 256 LibriSpeech-shaped utterances on three streams (bench.py's default): sampled utterances bit-identical to running them alone."""
    m, sd = _model("EfficientConformerCTCSmall", 0)
    enc = m.encoder
    lens = synth.libri_lengths(256, seed=1234)
    audio = torch.from_numpy(synth.make_audio(lens, seed=1234)).cuda()
    ln = torch.from_numpy(lens).cuda()
    enc.ragged, enc.sub_batches, enc.sub_batch_streams = True, 3, 3
    for _ in range(2):
        out, out_len, labels, label_len = m.encode_greedy(audio, ln, x_len_host=lens)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    enc.ragged, enc.sub_batches = False, 1
    for b in (0, 1, 79, 80, 127, 128, 200, 255):
        li = int(lens[b])
        alone, alone_len, _ = enc(audio[b:b + 1, :li].contiguous(), ln[b:b + 1].contiguous())
        tb = int(alone_len[0])
        assert int(out_len[b]) == tb and torch.equal(out[b, :tb], alone[0]) and float(out[b, tb:].abs().sum()) == 0.0, b
        _, lab, n = m._head(alone, alone_len)
        assert int(label_len[b]) == int(n[0]) and torch.equal(labels[b, :tb], lab[0, :tb])


# ------------------------------------------------------------------ streaming contexts / causal (SURVEY.md 8f-4 tail)
def _stream_model(gname, g):
    small = "small" in gname
    cfg = named_config("EfficientConformerCTCSmall" if small else "Tiny")
    extra = {k[4:]: (bool(g[k]) if k == "cfg/causal" else int(g[k])) for k in g.files if k.startswith("cfg/")}
    cfg["encoder_params"] = dict(cfg["encoder_params"], **extra)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, int(g["weight_seed"]), cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}
    return m.cuda(), osd, small


_STREAM = sorted(f for f in os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")) if f.startswith("stream_"))


@pytest.mark.parametrize("gname", _STREAM)
def test_streaming_and_causal_vs_reference_goldens(golden_dir, gname):
    """`causal` and finite `left_context` / `right_context` (reference encoders.py:68, 94; attentions.py:1377-1403, 506, 1243-1247;
    layers.py:97-101): band-masked attention with key-block skipping, causal relative tables, causal depthwise padding - against the
    reference encoder run with those settings (tools/make_goldens.py --only-streaming).  bf16 tolerance 0.06 max / 0.010 mean; every
    frame counts (pad frames included: their fully masked rows follow the reference's uniform softmax)."""
    g = np.load(os.path.join(golden_dir, gname))
    m, sd, small = _stream_model(gname, g)
    lens = g["mel_len"].tolist()
    mel, ln = synth.make_mel(len(lens), 80, max(lens), lens, seed=int(g["mel_seed"]))
    out, out_len, _ = m.encoder.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda())
    assert out_len.cpu().tolist() == g["out_len"].tolist()
    ref = torch.from_numpy(g["out_rows"] if small else g["out"])
    got = out.cpu()[:, ::4] if small else out.cpu()
    d = (got - ref).abs()
    print("%s: max %.4f mean %.5f" % (gname, float(d.max()), float(d.mean())))
    assert float(d.max()) < STREAM_MAX and float(d.mean()) < STREAM_MEAN
    if small:
        logits, _, _ = m._head(out, out_len, want_logits=True)
        am = logits.argmax(-1).cpu().numpy()
        valid = np.arange(am.shape[1])[None, :] < g["out_len"][:, None]
        safe = (g["margin"] > 0.15) & valid
        assert safe.sum() > 20 and np.array_equal(am[safe], g["argmax"][safe])
    again, _, _ = m.encoder.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda())
    assert torch.equal(again, out)


@pytest.mark.parametrize("extra", [dict(causal=True), dict(left_context=20, right_context=4), dict(causal=True, left_context=6)])
def test_streaming_ragged_batch_equals_utterances_alone_and_the_oracle(extra):
    cfg = named_config("Tiny")
    cfg["encoder_params"] = dict(cfg["encoder_params"], **extra)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 7, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}
    lens = np.array([48000, 41000, 30160, 22000, 12000, 3000], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=4))
    _ragged_vs_alone(m.cuda(), osd, audio, lens, 2, tol=(STREAM_MAX, STREAM_MEAN))


@pytest.mark.parametrize("exact", ["fp32", "split"])
@pytest.mark.parametrize("gname", [f for f in _STREAM if "_ctx_" in f])
def test_label_exact_modes_with_finite_contexts_vs_reference_goldens(golden_dir, gname, exact):
    """Round 4: finite left_context / right_context (reference attentions.py:1377-1403: ONE additive -1e9 on max(streaming mask, padding mask))
    in the two label-exact modes, against the reference run with those settings: encoder output within 2e-4 (every frame, pad frames included),
    per-frame argmax identical wherever the reference's top-2 margin exceeds 1e-3.  (`causal` stays on the bf16 path: next test.)"""
    g = np.load(os.path.join(golden_dir, gname))
    m, sd, small = _stream_model(gname, g)
    m.encoder.precision = exact
    lens = g["mel_len"].tolist()
    mel, ln = synth.make_mel(len(lens), 80, max(lens), lens, seed=int(g["mel_seed"]))
    out, out_len, _ = m.encoder.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda())
    assert out_len.cpu().tolist() == g["out_len"].tolist()
    ref = torch.from_numpy(g["out_rows"] if small else g["out"])
    got = out.cpu()[:, ::4] if small else out.cpu()
    d = (got - ref).abs()
    print("%s %s: max %.2e mean %.2e" % (gname, exact, float(d.max()), float(d.mean())))
    assert float(d.max()) < 2e-4
    if small:
        logits, _, _ = m._head(out, out_len, want_logits=True)
        am = logits.argmax(-1).cpu().numpy()
        valid = np.arange(am.shape[1])[None, :] < g["out_len"][:, None]
        safe = (g["margin"] > 1e-3) & valid
        assert safe.sum() > 20 and np.array_equal(am[safe], g["argmax"][safe])


def test_fp32_mode_rejects_causal_configurations():
    """The fp32-MFMA mode (csrc/exact.hip) has no causal kernels and says so; the split mode took them over in round 6 (tests/test_gpu_round6.py)."""
    cfg = named_config("Tiny")
    cfg["encoder_params"] = dict(cfg["encoder_params"], causal=True)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 7, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.encoder.precision = "fp32"
    m = m.cuda()
    mel, ln = synth.make_mel(2, 80, 100, [100, 77], seed=1)
    with pytest.raises(_lib.EffconfError, match="causal"):
        m.encoder.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda())


@pytest.mark.parametrize("name,nsub", [("ConformerCTCSmall", 1), ("ConformerCTCSmall", 2), ("ConformerCTCLarge", 3)])
def test_ragged_batch_with_the_two_layer_subsampler(name, nsub):
    """Plain Conformer configurations (two Conv2d subsampling layers, modules.py:232-249): both convolutions and the Linear run on the
    rectangular image - layer 1 zero-fills every utterance's image behind its own last frame, which is what layer 2 sees when the
    utterance runs alone - and the valid rows are gathered into the ragged row space.  Bit-identical to every utterance alone, within
    tolerance of the oracle; lengths chosen so that both layers' floor divisions differ between utterances."""
    m, sd = _model(name, 1)
    lens = np.array([48000, 47841, 40000, 30319, 22000, 12001, 3000], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=6))
    _ragged_vs_alone(m, sd, audio, lens, nsub, oracle=(name == "ConformerCTCSmall"))


def test_ragged_positional_cache_follows_the_longest_utterance():
    """Two ragged batches with the same batch size, row pitch and row TOTALS but different longest utterances: the cached positional
    projections (built for the longest utterance) must not be reused across them."""
    m, sd = _model("Tiny", 7)
    enc = m.encoder
    enc.ragged = True
    n = 48000
    la = np.array([48000, 16000], dtype=np.int64)
    lb = np.array([32000, 32000], dtype=np.int64)        # same total frames, shorter longest utterance
    audio = torch.from_numpy(synth.make_audio(np.array([n, n], dtype=np.int64), seed=8)).cuda()
    outs = {}
    for tag, l in (("a", la), ("b", lb), ("a2", la), ("b2", lb)):
        o, _, _ = enc(audio, torch.from_numpy(l).cuda(), x_len_host=l)
        outs[tag] = o.clone()
    assert torch.equal(outs["a"], outs["a2"]) and torch.equal(outs["b"], outs["b2"])
    enc.ragged = False
    for tag, l in (("a", la), ("b", lb)):
        for b in range(2):
            alone, al, _ = enc(audio[b:b + 1, :int(l[b])].contiguous(), torch.from_numpy(l[b:b + 1]).cuda())
            assert torch.equal(outs[tag][b, :int(al[0])], alone[0]), (tag, b)


# ------------------------------------------------------------------ fp32 mode: the tiled attention kernel == the one-wave-per-row kernel
@pytest.mark.parametrize("name,tm,lens", [
    ("Tiny", 100, [100, 77, 52, 0]),
    ("EfficientConformerCTCSmall", 1001, [1001, 640, 333]),         # grouped heads of 90 / 42 / 60 columns, Tg = 167 / 126 / 126
    ("EfficientConformerCTCLarge", 701, [701, 350]),                # an odd head width (135) and 3 output columns per lane
    ("ConformerCTCLarge", 501, [501, 77]),
])
def test_exact_mode_tiled_attention_is_bit_identical_to_the_row_kernel(name, tm, lens):
    """exact.hip: ex_attention2_kernel stages K / E / V blocks in LDS for 32 (16) query rows and keeps every sum in the order of
    ex_attention_kernel (attentions.py:549-718), so the whole fp32 forward must not change by one bit - including an empty utterance
    (all keys masked: the reference's uniform softmax) and a last query tile that is only partly valid."""
    m, _ = _model(name, 11, "fp32")
    mel, ln = synth.make_mel(len(lens), 80, tm, lens, seed=97)
    mel_d, ln_d = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
    m.encoder.set_option("exact_attention", 1)
    ref, ref_len, _ = m.encoder.forward_mel(mel_d, ln_d)
    for variant in (0, 2):                                          # 32-row (8 waves) and 16-row workgroups
        m.encoder.set_option("exact_attention", variant)
        out, out_len, _ = m.encoder.forward_mel(mel_d, ln_d)
        assert torch.equal(out_len, ref_len)
        assert torch.isfinite(out).all()
        assert torch.equal(out, ref), variant


# ------------------------------------------------------------------ batching front door: asynchronous path, ragged batches without zero-fill
def test_front_door_device_fn_path_with_stale_pad_samples_equals_the_list_path():
    """FrontDoor(device_fn=...) (no synchronisation inside the loop, ids downloaded asynchronously) must return what the list-returning
    path returns; with ragged batches and zero_pad = False the pad samples of the reused pinned staging buffer hold the PREVIOUS run's
    audio (here: 100x louder) and must not reach any output (utils/preprocessing.py:33-45 zero-pads; a ragged batch never reads them)."""
    from efficientconformer_amd import FrontDoor
    m, _ = _model("Tiny", 5)
    m.encoder.ragged = True
    g = torch.Generator().manual_seed(11)
    lens = [16000, 15999, 9000, 8000, 7777, 4000, 3999, 1600, 801, 800, 400, 16000, 12000]
    waves = [0.1 * torch.randn(n, generator=g) for n in lens]
    loud = [10.0 * torch.randn(16000, generator=g) for _ in lens]
    want = FrontDoor(m.greedy_labels, "cuda", max_batch=4).run(waves)
    door = FrontDoor(device_fn=lambda x, n, hl: m.encode_greedy(x, n, x_len_host=hl)[2:], device="cuda", max_batch=4, workers=3, zero_pad=False)
    door.run(loud)                                   # fills both staging buffers to the brim
    got = door.run(waves)
    assert got == want
    assert door.run(waves) == want                   # and again on the reused buffers
    assert sum(len(x) for x in want) > 0


# ------------------------------------------------------------------ attention maps (the third return value of the reference's forward)
@pytest.mark.parametrize("gname", ["tiny_T47.npz", "tiny_T100.npz"])
@pytest.mark.parametrize("precision", ["bf16", "fp32", "split"])
def test_attention_maps_vs_the_reference(golden_dir, gname, precision):
    """ConformerEncoder.forward(..., return_attentions=True) returns one (B, H, Tg, Tg) softmax map per block (reference
    encoders.py:126-142, attentions.py:620 / 718).  First / last block against the reference's own maps (tools/make_goldens.py: atts[0],
    atts[-1]), every block against the oracle's; rows sum to one; the forward's other outputs do not change.  bf16 path: the maps are
    recomputed in fp32 from bf16 Q / K / E (tolerance 0.02 on probabilities); fp32 path: 2e-5."""
    g = np.load(os.path.join(golden_dir, gname))
    m, sd = _model("Tiny", int(g["weight_seed"]), precision)
    b, tm = len(g["mel_len"]), int(g["mel_len"].max())
    mel, ln = synth.make_mel(b, 80, tm, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    mel_d, ln_d = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
    plain, plain_len, none = m.encoder.forward_mel(mel_d, ln_d)
    assert all(a is None for a in none)
    out, out_len, atts = m.encoder.forward_mel(mel_d, ln_d, return_attentions=True)
    # split mode (round 6): the forward with maps runs split.hip's scores-in-memory kernels (the maps are their by-product), the plain forward the fused
    # kernels of sxf.hip / sxf_ffn.hip - the same arithmetic in another summation order
    assert (torch.equal(out, plain) if precision != "split" else float((out - plain).abs().max()) < 2e-5) and torch.equal(out_len, plain_len)
    trace = {}
    with torch.no_grad():
        R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(ln), sd, m.encoder.plan, trace)
    tol = 0.02 if precision == "bf16" else 2e-5
    assert len(atts) == len(m.encoder.plan.blocks)
    worst = 0.0
    for k, a in enumerate(atts):
        ref = trace["blocks.%d.att_w" % k]
        assert tuple(a.shape) == tuple(ref.shape)
        assert float((a.sum(-1) - 1.0).abs().max()) < 1e-4
        worst = max(worst, float((a.cpu() - ref).abs().max()))
    print("attention maps, %s, %s: max |p - p_ref| over all blocks %.2e" % (gname, precision, worst))
    assert worst < tol
    assert float((atts[0].cpu() - torch.from_numpy(g["att0"])).abs().max()) < tol
    assert float((atts[-1].cpu() - torch.from_numpy(g["att_last"])).abs().max()) < tol
    again, _, none = m.encoder.forward_mel(mel_d, ln_d)            # the registration does not outlive the call
    assert torch.equal(again, plain) and all(a is None for a in none)


@pytest.mark.parametrize("extra", [dict(causal=True), dict(left_context=20, right_context=4), dict(causal=True, left_context=6)])
def test_attention_maps_with_streaming_contexts_vs_oracle(extra):
    """The maps of a streaming / causal encoder carry the band mask (attentions.py:1377-1403: streaming_mask.maximum(padding_mask),
    additive -1e9): masked entries are exactly zero, the rest matches the oracle; ragged batches return the padded rectangles."""
    cfg = named_config("Tiny")
    cfg["encoder_params"] = dict(cfg["encoder_params"], **extra)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 7, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}
    m = m.cuda()
    mel, ln = synth.make_mel(3, 80, 120, [120, 77, 30], seed=9)
    out, out_len, atts = m.encoder.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda(), return_attentions=True)
    trace = {}
    with torch.no_grad():
        R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(ln), osd, m.encoder.plan, trace)
    for k, a in enumerate(atts):
        ref = trace["blocks.%d.att_w" % k]
        assert float((a.cpu() - ref).abs().max()) < 0.02, k
        assert bool((a.cpu()[ref == 0] == 0).all())
    # ragged batch (round 4: no longer refused): every utterance's own Tg x Tg block of the padded rectangle is the map of that utterance
    # run ALONE (rectangular path, B = 1) - bit for bit, band mask included - and everything outside the block is zero
    m.encoder.ragged = True
    rout, rlen, ratts = m.encoder.forward_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda(), return_attentions=True, x_len_host=ln)
    m.encoder.ragged = False
    for b in range(3):
        tb = int(ln[b])
        _, _, alone = m.encoder.forward_mel(torch.from_numpy(mel[b:b + 1, :, :tb].copy()).cuda(), torch.from_numpy(ln[b:b + 1]).cuda(), return_attentions=True)
        for k, (ra, al) in enumerate(zip(ratts, alone)):
            tg = al.shape[-1]
            assert torch.equal(ra[b, :, :tg, :tg], al[0]), (b, k, float((ra[b, :, :tg, :tg] - al[0]).abs().max()))
            assert float(ra[b, :, tg:, :].abs().sum()) == 0.0 and float(ra[b, :, :, tg:].abs().sum()) == 0.0


# ------------------------------------------------------------------ CTC head with split-bf16 operands (the bf16 path's default)
@pytest.mark.parametrize("name,vocab", [("EfficientConformerCTCSmall", 256), ("EfficientConformerCTCSmall", 1000), ("EfficientConformerCTCMedium", 256),
                                        ("EfficientConformerCTCLarge", 256), ("Tiny", 48)])
def test_ctc_head_split_bf16_vs_the_fp32_head(name, vocab):
    """ctc_argmax_bf16x3_kernel (x_hi W_hi + x_hi W_lo + x_lo W_hi on the bf16 MFMA, fp32 accumulation; model_ctc.py:49, 96-99) against the
    fp32 head: logits within 2^-15 of the operand magnitudes (measured ~1e-5 relative), argmax identical wherever the fp32 top-2 margin
    exceeds 1e-3 - on frame counts that are no multiple of the row tile, D = 240 / 360 (not a multiple of 16) / 720, one and four column
    passes.  The fp32-operand mode keeps the fp32 head."""
    cfg = named_config(name)
    m = ModelCTC(cfg["encoder_params"], {"vocab_size": vocab})
    sd = synth.make_state_dict(m.encoder.plan, 5, vocab, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    m.encoder._ensure_packed()
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(3, 77, m.encoder.plan.dim_out, generator=g).cuda()
    ln = torch.tensor([77, 50, 9]).cuda()
    m.encoder.set_option("ctc_mfma", 1)
    l1, lab1, n1 = m._head(enc, ln, want_logits=True)
    m.encoder.set_option("ctc_mfma", 2)
    l2, lab2, n2 = m._head(enc, ln, want_logits=True)
    scale = float(l1.abs().max())
    err = float((l2 - l1).abs().max())
    print("%s V=%d: split-bf16 head vs fp32 head: max |dlogit| %.2e (logits up to %.1f)" % (name, vocab, err, scale))
    assert err < 3e-5 * max(scale, 1.0) * 8
    top = l1.topk(2, dim=-1).values
    safe = (top[..., 0] - top[..., 1]) > 1e-3
    assert bool(safe.float().mean() > 0.9)
    assert torch.equal(l2.argmax(-1)[safe], l1.argmax(-1)[safe])
    _, lab3, n3 = m._head(enc, ln)                                     # without the logits output
    assert torch.equal(lab3, lab2) and torch.equal(n3, n2)


def test_fp32_mode_keeps_the_fp32_ctc_head():
    """The label-exact mode (precision = "fp32") runs the fp32 head whatever ctc_mfma says: its logits equal ctc_mfma = 1's bit for bit."""
    m, _ = _model("Tiny", 5, "fp32")
    m.encoder._ensure_packed()
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(3, 40, m.encoder.plan.dim_out, generator=g).cuda()
    ln = torch.tensor([40, 33, 9]).cuda()
    m.encoder.set_option("ctc_mfma", 1)
    l1, lab1, _ = m._head(enc, ln, want_logits=True)
    m.encoder.set_option("ctc_mfma", 2)
    l2, lab2, _ = m._head(enc, ln, want_logits=True)
    assert torch.equal(l2, l1) and torch.equal(lab2, lab1)
