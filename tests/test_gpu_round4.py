"""-m gpu tests added in round 4: advisor findings of round 3 (split-bf16 CTC head at vocabulary sizes that are no multiple of 256,
ragged row ranges under ShardedEncoder, host / device length consistency), the mode matrix (fp32 / split precision x ragged x
streaming x attention maps) and the kernels rewritten this round.  Tolerances as tests/test_gpu_round3.py."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from efficientconformer_amd import ModelCTC, _lib, named_config, synth
from oracle import ref_encoder as R

pytestmark = pytest.mark.gpu


def _model(name, seed, precision="bf16", vocab=None):
    cfg = named_config(name)
    v = vocab or cfg["tokenizer_params"]["vocab_size"]
    m = ModelCTC(cfg["encoder_params"], {"vocab_size": v})
    sd = synth.make_state_dict(m.encoder.plan, seed, v, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v_) for k, v_ in sd.items()})
    m.encoder.precision = precision
    osd = {k[len("encoder."):] if k.startswith("encoder.") else k: v_ for k, v_ in sd.items()}
    return m.cuda(), osd


# ------------------------------------------------------------------ advisor (round 3, medium): split-bf16 head, V % 256 != 0
@pytest.mark.parametrize("name,vocab", [("EfficientConformerCTCSmall", 128), ("EfficientConformerCTCSmall", 300), ("Tiny", 5000),
                                        ("EfficientConformerCTCLarge", 257), ("Tiny", 65)])
def test_split_bf16_head_reads_inside_its_weight_images_for_any_vocabulary(name, vocab, monkeypatch):
    """ctc_argmax_bf16x3_kernel covers 256 columns per pass (4 waves x 64); the packed fc images were padded to 64 columns only, so for
    V = 128 / 300 / 5000 the waves beyond round_up(V, 64) read past the buffer (advisor, round 3).  The images are padded to whole passes
    now: with NaN guard regions around every parameter buffer (EFFCONF_POISON_GUARDS) the logits are finite, identical to the run without
    guards and within the split head's bound of the fp32 head (model_ctc.py:49, 96-99)."""
    g = torch.Generator().manual_seed(11)
    res = {}
    for guards in ("0", "1"):
        monkeypatch.setenv("EFFCONF_POISON_GUARDS", guards)
        m, _ = _model(name, 5, vocab=vocab)
        m.encoder._ensure_packed()
        enc = torch.randn(3, 70, m.encoder.plan.dim_out, generator=torch.Generator().manual_seed(3)).cuda()
        ln = torch.tensor([70, 41, 7]).cuda()
        m.encoder.set_option("ctc_mfma", 2)
        l2, lab2, n2 = m._head(enc, ln, want_logits=True)
        m.encoder.set_option("ctc_mfma", 1)
        l1, lab1, n1 = m._head(enc, ln, want_logits=True)
        assert torch.isfinite(l2).all(), (name, vocab, guards)
        res[guards] = (l2.cpu(), lab2.cpu(), n2.cpu())
        scale = max(float(l1.abs().max()), 1.0)
        assert float((l2 - l1).abs().max()) < 2.4e-4 * scale
        top = l1.topk(2, dim=-1).values
        safe = (top[..., 0] - top[..., 1]) > 1e-3
        assert torch.equal(l2.argmax(-1)[safe], l1.argmax(-1)[safe])
        del m
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][2], res["1"][2])


# ------------------------------------------------------------------ split-precision GEMM kernel alone (csrc/split.hip)
@pytest.mark.parametrize("m,n,k,epi", [(333, 308, 52, 1), (1000, 100, 100, 0), (129, 4, 4, 2), (128, 128, 32, 0), (4097, 360, 120, 0), (2500, 120, 480, 2),
                                       (777, 672, 168, 1), (64, 2880, 720, 1)])
def test_split_gemm_kernel_alone_vs_float64(m, n, k, epi):
    """sx_gemm_kernel through effconf_debug_sx_gemm: c = epilogue(a w^T + b) with the weight given as its two fp16 images (h = fp16(w),
    l = fp16((w - h) * 2048)) against a float64 product - M / N / K tails (K % 32 != 0: zero-filled k-tile, N % 128, M % 128), one k-tile,
    every epilogue (plain, Swish, residual).  Products are accurate to ~2^-21: relative error of the result below 2e-6 (measured 1 - 8e-7);
    nothing outside the [m][n] block is written."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sx_gemm_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "sx_gemm_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = _lib.load_debug()
    g = np.random.default_rng(7 * m + n + k)
    a = g.standard_normal((m, k), dtype=np.float32)
    w = (g.standard_normal((n, k), dtype=np.float32) / np.sqrt(k)).astype(np.float32)
    bias = (0.1 * g.standard_normal(n)).astype(np.float32)
    if epi == 1:                 # outlier pre-activations (advisor, round 4): Swish(-100) .. Swish(-2000) must be 0 (-x e^x underflows), not NaN from exp2 overflow
        bias[: min(n, 6)] = np.array([-100.0, -89.0, -88.0, -2000.0, 95.0, -1e4], dtype=np.float32)[: min(n, 6)]
    r = g.standard_normal((m, n), dtype=np.float32)
    hi, lo, ldh = mod.split_images(w)
    ad, hd, ld, bd, rd = (torch.from_numpy(x).cuda() for x in (a, hi.view(np.int16), lo.view(np.int16), bias, r))
    ldc = n + 8
    c = torch.full((m + 3, ldc), 7.0, device="cuda")            # guard rows / columns: must stay untouched
    _lib.check(lib.effconf_debug_sx_gemm(ad.data_ptr(), k, hd.data_ptr(), ld.data_ptr(), ldh, bd.data_ptr(), m, n, k, epi, c.data_ptr(), ldc,
                                         rd.data_ptr(), n, C.c_float(0.5), torch.cuda.current_stream().cuda_stream), "sx_gemm")
    torch.cuda.synchronize()
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if epi == 1:
        with np.errstate(over="ignore"):
            ref = ref / (1.0 + np.exp(-ref))
    elif epi == 2:
        ref = r + 0.5 * ref
    got = c.cpu().numpy()
    err = float(np.abs(got[:m, :n] - ref).max() / max(np.abs(ref).max(), 1.0))
    print("sx_gemm %dx%dx%d epi %d: rel err %.2e" % (m, n, k, epi, err))
    assert np.isfinite(got[:m, :n]).all()
    assert err < 2e-6
    assert np.all(got[m:] == 7.0) and np.all(got[:, n:] == 7.0)


# ------------------------------------------------------------------ CTC head on bf16 rows (gathered chunks of the multi-rank path, bf16 wire)
@pytest.mark.parametrize("name,vocab", [("EfficientConformerCTCSmall", 256), ("EfficientConformerCTCSmall", 300), ("EfficientConformerCTCLarge", 1000), ("Tiny", 65)])
def test_ctc_head_on_bf16_rows_is_bit_identical_to_the_fp32_input_head_on_the_widened_rows(name, vocab):
    """effconf_ctc_greedy_bf16 (include/effconf.h): bf16 rows are their own hi half, the lo half is zero - the kernel drops the x_lo W_hi MFMAs and
    the widening pass; logits, label ids and label counts equal those of effconf_ctc_greedy on the same rows widened to fp32, bit for bit
    (reference: the head after the all-gather, main.py:217-220 + model_ctc.py:90-133).  Exact modes refuse bf16 rows by widening them in
    ModelCTC._head (they keep the fp32 head)."""
    m, _ = _model(name, 9, vocab=vocab)
    m.encoder._ensure_packed()
    enc = torch.randn(5, 83, m.encoder.plan.dim_out, generator=torch.Generator().manual_seed(4)).cuda().to(torch.bfloat16)
    ln = torch.tensor([83, 60, 33, 1, 0]).cuda()
    l_b, lab_b, n_b = m._head(enc, ln, want_logits=True)
    l_f, lab_f, n_f = m._head(enc.float(), ln, want_logits=True)
    assert torch.equal(l_b, l_f) and torch.equal(n_b, n_f)
    for b in range(5):
        assert torch.equal(lab_b[b, :int(n_b[b])], lab_f[b, :int(n_f[b])])
    _, lab_b2, n_b2 = m._head(enc, ln)                              # the label-only launch (no logits buffer)
    assert torch.equal(n_b2, n_b) and all(torch.equal(lab_b2[b, :int(n_b[b])], lab_b[b, :int(n_b[b])]) for b in range(5))
    lib = _lib.load()
    m.encoder.precision = "fp32"
    m.encoder._ensure_packed()
    ws = torch.empty(5 * 83 * 4, dtype=torch.uint8, device="cuda")
    rc = lib.effconf_ctc_greedy_bf16(m.encoder._handle, enc.data_ptr(), ln.data_ptr(), 5, 83, lab_b.data_ptr(), n_b.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"bf16 path" in lib.effconf_last_error()


# ------------------------------------------------------------------ small launches: 2-wave chain workgroups (option chain_small_m)
@pytest.mark.parametrize("name,batch,seconds", [("EfficientConformerCTCSmall", 4, 10.0), ("EfficientConformerCTCSmall", 1, 3.3), ("EfficientConformerCTCMedium", 3, 6.0),
                                                 ("EfficientConformerTransducerSmall", 2, 5.0)])
def test_small_launches_on_two_wave_chain_workgroups_are_bit_identical(name, batch, seconds):
    """Chain launches of at most chain_small_m rows (default 4096) run as 2-wave workgroups, 64 rows each, instead of 8-wave ones (csrc/chain.hip,
    launch_chain_kind: small-batch latency).  A wave computes its 32 rows with the same instruction sequence in both shapes, so the encoder output
    is bit-identical with the option off (chain_small_m = 0: the wide shapes for every launch) - rectangular and ragged batches."""
    cfg = named_config(name)
    from efficientconformer_amd import Transducer
    m = (Transducer if cfg["model_type"] == "Transducer" else ModelCTC).from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 3, None, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.cuda()
    lens = np.array([int(16000 * seconds * (1.0 - 0.13 * i)) for i in range(batch)], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=5)).cuda()
    ln = torch.from_numpy(lens).cuda()
    outs = {}
    for small in (4096, 0):
        m.encoder.set_option("chain_small_m", small)
        for ragged in (False, True):
            m.encoder.ragged = ragged
            kw = {"x_len_host": lens} if ragged else {}
            enc, el, _ = m.encoder(audio, ln, **kw)
            outs[(small, ragged)] = (enc.clone(), el.clone())
    for ragged in (False, True):
        a, b = outs[(4096, ragged)], outs[(0, ragged)]
        assert torch.equal(a[1], b[1])
        assert torch.isfinite(a[0].float()).all()
        assert torch.equal(a[0], b[0]), (name, ragged, float((a[0].float() - b[0].float()).abs().max()))
