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
