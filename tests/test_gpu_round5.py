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


# ------------------------------------------------------------------ the rectangular path as alternating range shapes on one stream
def test_alternating_rectangular_range_shapes_on_one_stream_cost_what_each_shape_costs_alone():
    """Two kept profiles of round 4 (profiles/r4_03_bench_ragged0.json, r4_04_bench_ragged0.json) show the roofline leg of `bench.py --ragged 0` - the step's
    three trimmed row ranges as three forwards of different shapes, one after the other on one stream - with the class gemm_other (chain B + projections) at
    14.6 / 17.1 ms per step in a 5.9 ms step; r4_02, three commits earlier, at 1.8 ms.  It does not reproduce (round 5: 1.79 - 1.83 ms on this tree AND on round
    4's kernel selection, profiles/r5_20_*; the records came from tools/gpu_evidence.sh sessions) - this test is the guard the verdict asked for: per class, the
    library's own launch brackets (effconf_profile_*) of shapes A, B, C alternating cost at most 1.5 x what the same forwards cost with every shape repeated
    back to back (positional-embedding cache misses of the alternation included: its projections are part of gemm_other)."""
    import ctypes as C
    from efficientconformer_amd import _lib
    m = _any_model("EfficientConformerCTCSmall", 3)
    lib = _lib.load()
    rng = np.random.default_rng(5)
    shapes = [(24, 15.2), (40, 11.0), (56, 7.4)]                      # rows x seconds: the bench's three trimmed ranges in miniature
    batches = []
    for b, sec in shapes:
        lens = np.sort((16000 * sec * (1.0 - 0.25 * rng.random(b))).astype(np.int64))[::-1].copy()
        lens[0] = int(16000 * sec)
        batches.append((torch.from_numpy(synth.make_audio(lens, seed=b)).cuda(), torch.from_numpy(lens).cuda()))
    enc = m.encoder
    enc.ragged, enc.sub_batches = False, 1
    enc._ensure_packed()
    h = enc._handle
    names = ["mel", "subsample_conv", "gemm_ffn", "gemm_other", "layernorm", "attention", "dwconv", "misc"]

    def run(order, reps):
        for a, l in batches:                                          # un-profiled pass: workspaces, caches
            enc(a, l)
        torch.cuda.synchronize()
        _lib.check(lib.effconf_profile_enable(h, 1), "profile_enable")
        for _ in range(reps):
            for i in order:
                enc(*batches[i])
        torch.cuda.synchronize()
        out = {}
        for ci, cname in enumerate(names):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(lib.effconf_profile_read(h, ci, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "profile_read")
            out[cname] = (ms.value, n.value)
        _lib.check(lib.effconf_profile_enable(h, 0), "profile_enable")
        return out
    alone = {k: 0.0 for k in names}
    for i in range(3):
        r = run([i], 4)
        for k in names:
            alone[k] += r[k][0]
    mixed = run([0, 1, 2], 4)
    for k in names:
        assert mixed[k][0] <= 1.5 * alone[k] + 0.2, (k, mixed[k], alone[k])          # ms over 12 forwards each way; + 0.2 ms of slack for the small classes


# ------------------------------------------------------------------ the wave-parallel utterance search beyond one 256-entry chunk
@pytest.mark.parametrize("name,batch", [("Tiny", 300), ("Tiny", 700), ("EfficientConformerCTCSmall", 520)])
def test_ragged_launches_of_more_than_256_utterances(name, batch):
    """`ragged_find_wave` (common.h; the head of every attention and depthwise-convolution workgroup of a ragged launch) counts the prefix sums <= the
    workgroup id with ballots, 256 entries per pass: batches of more than 256 (two passes) and 512 (three) utterances in ONE row range, some of them one
    workgroup long, against sampled utterances run alone - bit-identical."""
    m = _any_model(name, 11)
    rng = np.random.default_rng(batch)
    lens = np.sort((16000 * (0.05 + 2.4 * rng.random(batch))).astype(np.int64))[::-1].copy()
    lens[-3:] = (400, 320, 260)                                       # down to the shortest legal utterance (n_fft / 2 < samples)
    audio = torch.from_numpy(synth.make_audio(lens, seed=9)).cuda()
    ln = torch.from_numpy(lens).cuda()
    enc = m.encoder
    enc.ragged, enc.sub_batches = True, 1
    out, out_len, _ = enc(audio, ln, x_len_host=lens)
    assert torch.isfinite(out.float()).all()
    enc.ragged = False
    for b in (0, 1, 255, 256, 257, batch // 2, batch - 2, batch - 1):
        li = int(lens[b])
        alone, alone_len, _ = enc(audio[b:b + 1, :li].contiguous(), ln[b:b + 1].contiguous())
        tb = int(alone_len[0])
        assert int(out_len[b]) == tb and torch.equal(out[b, :tb], alone[0]) and float(out[b, tb:].float().abs().sum()) == 0.0, b


# ------------------------------------------------------------------ depthwise convolution on the matrix pipe
@pytest.mark.parametrize("name,batch,seconds", [("EfficientConformerCTCSmall", 4, 7.3), ("ConformerCTCSmall", 3, 4.0), ("EfficientConformerCTCMedium", 2, 3.1),
                                                 ("EfficientConformerTransducerSmall", 2, 5.0)])
def test_depthwise_conv_on_the_matrix_pipe_matches_the_valu_kernel(name, batch, seconds):
    """dwconv_mfma_kernel (conv.hip; the stride-1 layers, kernel sizes 15 and 31) computes every channel's convolution as 4 x 4 x 4 Toeplitz products on
    v_mfma_f32_4x4x4_16b_bf16 with the fp32 folded taps split into bf16 hi + lo halves: the same products as dwconv_kernel's fp32 FMAs up to 2^-17 of a tap and
    the summation order - a bf16 output may move by one ulp, which the following blocks spread to the bf16 noise floor of the encoder output.  Rectangular and ragged batches (the ragged one bit-identical to itself run alone is covered by the
    ragged tests, which run on the default = this kernel), channel tiles with pad channels (120 = 64 + 56, 176 = 2 x 64 + 48), tiles past an utterance's end."""
    m = _any_model(name, 3)
    lens = np.array([int(16000 * seconds * (1.0 - 0.21 * i)) for i in range(batch)], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=5)).cuda()
    ln = torch.from_numpy(lens).cuda()
    outs = {}
    for opt in (0, 2):                                                # 2: every supported kernel size (the default, 1, keeps size 31 on the VALU kernel)
        m.encoder.set_option("dwconv_mfma", opt)
        for ragged in (False, True):
            m.encoder.ragged = ragged
            kw = {"x_len_host": lens} if ragged else {}
            enc, el, _ = m.encoder(audio, ln, **kw)
            outs[(opt, ragged)] = (enc.float().clone(), el.clone())
    for ragged in (False, True):
        a, b = outs[(0, ragged)], outs[(2, ragged)]
        assert torch.equal(a[1], b[1]) and torch.isfinite(b[0]).all()
        d = (a[0] - b[0]).abs()
        # two bf16 paths of equal accuracy differ by about a third of their distance to the fp32 oracle (measured 0.013 - 0.016 max, 0.0023 - 0.0026 mean;
        # against the oracle both kernels: profiles/r5_35_dw_accuracy.txt)
        assert float(d.max()) < 0.04 and float(d.mean()) < 4e-3, (name, ragged, float(d.max()), float(d.mean()))


# ------------------------------------------------------------------ host lengths that disagree with the device lengths
def test_ragged_forward_refuses_host_lengths_that_differ_from_the_device_lengths():
    """`check_host_lengths` is True by default since round 5: grids and the workspace of a ragged forward are sized from `x_len_host`, the kernels index with the
    device `x_len` - a caller whose copies disagree gets a ValueError instead of an out-of-bounds access.  The comparison is remembered per (device tensor,
    version, host values): an in-place change of the device lengths is seen, and opting out (False) restores round 4's behaviour for callers that vouch for them."""
    m = _any_model("Tiny", 2)
    enc = m.encoder
    lens = np.array([30000, 24000, 17000, 9000], dtype=np.int64)
    audio = torch.from_numpy(synth.make_audio(lens, seed=3)).cuda()
    ln = torch.from_numpy(lens).cuda()
    enc.ragged, enc.sub_batches = True, 1
    assert enc.check_host_lengths is True
    out, out_len, _ = enc(audio, ln, x_len_host=lens)
    wrong = lens.copy(); wrong[2] -= 160
    with pytest.raises(ValueError):
        enc(audio, ln, x_len_host=wrong)
    out2, _, _ = enc(audio, ln, x_len_host=lens)                      # the memo holds the verified pair, a wrong one in between did not replace it
    assert torch.equal(out, out2)
    ln[2] -= 160                                                      # in-place change of the device tensor: its version moves, the pair is compared again
    with pytest.raises(ValueError):
        enc(audio, ln, x_len_host=lens)
    out3, out_len3, _ = enc(audio, ln, x_len_host=wrong)              # now `wrong` is right
    assert int(out_len3[2]) <= int(out_len[2]) and torch.isfinite(out3.float()).all()
    # a converted temporary (int32 x_len -> a fresh int64 tensor per call) or a freed-and-reallocated length tensor comes back from the
    # caching allocator at the SAME address with version 0 - the memo may not treat that as "the tensor I verified".  Same address, different lengths, stale
    # host copy: must raise on every call, in both forms.
    l32 = torch.from_numpy(wrong.astype(np.int32)).cuda()
    enc(audio, l32, x_len_host=wrong)                                 # verified through a temporary: nothing may be remembered for it
    l32b = torch.from_numpy(lens.astype(np.int32)).cuda()
    with pytest.raises(ValueError):
        enc(audio, l32b, x_len_host=wrong)
    fresh = torch.from_numpy(wrong).cuda()
    enc(audio, fresh, x_len_host=wrong)
    ptr = fresh.data_ptr()
    del fresh
    again = torch.from_numpy(lens).cuda()                             # the memo holds a reference to `fresh`, so its block cannot be recycled: a new address
    assert again.data_ptr() != ptr
    with pytest.raises(ValueError):
        enc(audio, again, x_len_host=wrong)
