"""-m gpu: the tiled bf16 GEMM kernels alone (csrc/gemm.hip, csrc/gemm256.hip) through the C ABI entry effconf_debug_gemm, against a
plain PyTorch fp32 reference of the same op on the same bf16-rounded operands (reference layers: models/layers.py:57-67 Linear,
modules.py:378-382 FFN, modules.py:502-508 pointwise conv + GLU).

Tolerances: fp32 outputs 2e-3 of the output magnitude (fp32 accumulation in a different order); bf16 outputs one bf16 ulp of the
output magnitude (2^-7) on top of that.  Shapes cover M / N / K tails, K not a multiple of 64 (the zero-chunk DMA tail of
gemm256.hip), the widths of the Large configurations (512, 720, 2048, 2880) and every tile choice."""
import ctypes as C

import pytest
import torch

from efficientconformer_amd import _lib

pytestmark = pytest.mark.gpu

F32, BF16, SWISH, RESID, GLU = 0, 1, 2, 3, 4


def _ru(x, m):
    return (x + m - 1) // m * m


def _bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _case(M, N, K, epi, wide, seed=0, lda_pad=0):
    """N counts output columns of the Linear (for GLU: N = 2 * channels, channels % 32 == 0 here)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    a16, w16 = a.to(torch.bfloat16), w.to(torch.bfloat16)
    ref = a16.float() @ w16.float().t() + bias
    lda = _ru(K, 8) + lda_pad
    a_dev = torch.full((M, lda), float("nan"), dtype=torch.bfloat16, device="cuda")   # columns >= K must never matter
    a_dev[:, :K] = a16
    Np, Kp = _ru(N, 128), _ru(K, 64)
    w_rows = w16
    b_rows = bias
    if epi == GLU:
        ch = N // 2
        assert ch % 32 == 0
        idx = torch.arange(N, device="cuda")
        blk, r = idx // 64, idx % 64
        src = torch.where(r < 32, blk * 32 + r, ch + blk * 32 + (r - 32))      # packed row -> reference row (a | b per 32 channels)
        w_rows, b_rows = w16[src], bias[src]
    w_dev = torch.zeros(Np, Kp, dtype=torch.bfloat16, device="cuda")
    w_dev[:N, :K] = w_rows
    b_dev = torch.zeros(Np, device="cuda")
    b_dev[:N] = b_rows
    r_dev = None
    alpha = 1.0
    if epi == F32:
        ldc = N
        c = torch.full((M, ldc), float("nan"), device="cuda")
        want = ref
    elif epi in (BF16, SWISH):
        ldc = _ru(N, 8)
        c = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device="cuda")
        want = ref if epi == BF16 else ref * torch.sigmoid(ref)
    elif epi == RESID:
        ldc = N
        alpha = 0.5
        r_dev = torch.randn(M, N, generator=g).cuda()
        c = torch.full((M, ldc), float("nan"), device="cuda")
        want = r_dev + alpha * ref
    else:
        ch = N // 2
        ldc = _ru(ch, 8)
        c = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device="cuda")
        want = ref[:, :ch] * torch.sigmoid(ref[:, ch:])
    lib = _lib.load_debug()
    rc = lib.effconf_debug_gemm(_ptr(a_dev), lda, _ptr(w_dev), Kp, _ptr(b_dev), M, N, K, epi, wide, _ptr(c), ldc,
                                _ptr(r_dev) if r_dev is not None else None, N, C.c_float(alpha), None)
    _lib.check(rc, "effconf_debug_gemm")
    torch.cuda.synchronize()
    got = c[:, :want.shape[1]].float()
    scale = float(want.abs().max())
    err = float((got - want).abs().max()) / scale
    tol = 2e-3 + (2.0 ** -7 if c.dtype == torch.bfloat16 else 0.0)
    assert torch.isfinite(got).all(), (M, N, K, epi, wide)
    assert err < tol, (M, N, K, epi, wide, err)
    if c.dtype == torch.bfloat16 and ldc > want.shape[1]:
        assert (c[:, want.shape[1]:].float() == 0).all()        # pad columns of a bf16 row buffer are written as zeros
    return got


SHAPES = [(300, 512, 512), (1000, 2048, 512), (513, 720, 2880), (777, 2880, 720), (256, 256, 64), (1, 128, 360), (2049, 360, 1440),
          (640, 1536, 512), (333, 1024, 200)]


@pytest.mark.parametrize("wide", [1, 2, 3])
@pytest.mark.parametrize("epi", [F32, BF16, SWISH, RESID])
@pytest.mark.parametrize("shape", SHAPES)
def test_linear_epilogues_vs_fp32_reference(shape, epi, wide):
    M, N, K = shape
    _case(M, N, K, epi, wide, seed=M + N + K)


@pytest.mark.parametrize("wide", [1, 2, 3])
@pytest.mark.parametrize("shape", [(300, 1024, 512), (1000, 1408, 720), (129, 128, 360), (2000, 2048, 1024)])
def test_glu_epilogue_vs_fp32_reference(shape, wide):
    M, N, K = shape
    _case(M, N, K, GLU, wide, seed=N)


def test_tile_choices_agree_bit_for_bit_on_fp32_outputs():
    """Every tile shape accumulates a k-ordered chain of the same MFMA over the same 16-wide k-steps: identical fp32 results."""
    outs = [_case(1000, 2048, 512, F32, wide, seed=3) for wide in (1, 2, 3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_operand_row_pitch_larger_than_k_and_poisoned_tail_columns():
    """lda > K with NaN in the columns past K: the K tail of the last k-tile must come from the zero chunk, never from the row."""
    for wide in (1, 2, 3):
        _case(500, 512, 360, F32, wide, seed=9, lda_pad=24)
        _case(500, 512, 424, BF16, wide, seed=10, lda_pad=8)


@pytest.mark.parametrize("wide", [2, 3])
@pytest.mark.parametrize("name,tm", [("EfficientConformerCTCLarge", 1001), ("ConformerCTCLarge", 501), ("EfficientConformerCTCMedium", 1001)])
def test_wide_configurations_end_to_end_on_the_lds_dma_gemm(golden_dir, name, tm, wide):
    """The whole encoder with every tiled GEMM layer forced onto gemm256.hip (by shape it is only picked at bench-size row counts),
    including the 257 .. 384-wide layers that otherwise run on the row-stationary kernels (LayerNorm as its own kernel then): same
    tolerance against the reference goldens as the default path; where only the tile differs (ConformerCTC-Large: every layer is
    tiled either way) the result is bit-identical to the 128 x 128 kernel's."""
    import os

    import numpy as np

    from efficientconformer_amd import ModelCTC, named_config, synth
    g = np.load(os.path.join(golden_dir, name + "_B2.npz"))
    m = ModelCTC(named_config(name)["encoder_params"], {"vocab_size": 256})
    sd = synth.make_state_dict(m.encoder.plan, int(g["weight_seed"]), 256, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    mel, ln = synth.make_mel(2, 80, tm, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    mel_d, ln_d = torch.from_numpy(mel).cuda(), torch.from_numpy(ln).cuda()
    m.encoder.set_option("wide_gemm", 1)
    base = m.encoder.forward_mel(mel_d, ln_d)[0].cpu()
    m.encoder.set_option("wide_gemm", wide)
    out, out_len, _ = m.encoder.forward_mel(mel_d, ln_d)
    assert out_len.cpu().tolist() == g["out_len"].tolist()
    d = (out[:, ::8].cpu().double() - torch.from_numpy(g["out_rows"]).double()).abs()
    print("%s wide %d: err max %.4f mean %.5f; max |wide - 128x128| %.3g" % (name, wide, float(d.max()), float(d.mean()), float((out.cpu() - base).abs().max())))
    assert float(d.max()) < 0.06 and float(d.mean()) < 0.010
    if name == "ConformerCTCLarge":
        assert torch.equal(out.cpu(), base)
    else:
        assert float((out.cpu() - base).abs().max()) < 0.08
