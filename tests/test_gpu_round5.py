"""-m gpu tests added in round 5: the column-pair chain kernels (csrc/chain2.hip) against chain.hip's, bit for bit."""
import numpy as np
import pytest
import torch

from efficientconformer_amd import ModelCTC, named_config, synth

pytestmark = pytest.mark.gpu


def _any_model(name, seed):
    cfg = named_config(name)
    from efficientconformer_amd import Transducer
    m = (Transducer if cfg["model_type"] == "Transducer" else ModelCTC).from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, seed, None, prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.cuda()


# widths: CTC-Small 120 / 168 / 240 (KS = 8 / 12 / 16), CTC-Medium 180 / 256 / 360 (12 / 16 / per-GEMM kernels; D % 8 == 4 at 180: the 8-byte
# Q/K/V stores), Transducer-Small 144 / 200 / 280? (padded widths 192 / 256), ConformerCTC-Small 176 (12)
@pytest.mark.parametrize("name,batch,seconds", [("EfficientConformerCTCSmall", 5, 9.0), ("EfficientConformerCTCSmall", 1, 2.1), ("EfficientConformerCTCMedium", 3, 6.0),
                                                 ("EfficientConformerTransducerSmall", 2, 5.0), ("ConformerCTCSmall", 3, 4.0)])
def test_column_pair_chains_are_bit_identical_to_the_single_wave_chains(name, batch, seconds):
    """chain2.hip runs the chains of the wide stages with a PAIR of waves per 32 rows (column halves; 8-wave workgroups, two waves per SIMD).  Every
    accumulator sees chain.hip's operations in chain.hip's order (first-GEMM k order, second-GEMM chunk order, LayerNorm sums continued across the
    pair), so the encoder output is bit-identical with the option off - burst and hooked refills, tail and head of chain A as one kernel
    (chain_full_max = 256) or two, rectangular and ragged batches, launches that are no multiple of 128 rows."""
    m = _any_model(name, 3)
    lens = np.array([int(16000 * seconds * (1.0 - 0.17 * i)) for i in range(batch)], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=5)).cuda()
    ln = torch.from_numpy(lens).cuda()
    m.encoder.set_option("chain_small_m", 0)           # chain.hip's wide shapes as the reference for every launch
    m.encoder.set_option("chain_pair_min_d", 0)        # the pair kernels wherever they exist (default: padded width 256 only)
    outs = {}
    variants = ((1, 192), (2, 192), (3, 256), (4, 192), (4, 256), (5, 256))      # (refill mode, second FFN weights row-major (192) / chunk-major (256))
    for pair, full in ((0, 192),) + variants:
        m.encoder.set_option("chain_w2cm", 1 if full == 256 else 0)
        m.encoder.set_option("chain_pair", pair)
        for ragged in (False, True):
            m.encoder.ragged = ragged
            kw = {"x_len_host": lens} if ragged else {}
            enc, el, _ = m.encoder(audio, ln, **kw)
            outs[(pair, full, ragged)] = (enc.clone(), el.clone())
    for ragged in (False, True):
        a = outs[(0, 192, ragged)]
        assert torch.isfinite(a[0].float()).all()
        for key in variants:
            b = outs[key + (ragged,)]
            assert torch.equal(a[1], b[1])
            assert torch.equal(a[0], b[0]), (name, key, ragged, float((a[0].float() - b[0].float()).abs().max()))
