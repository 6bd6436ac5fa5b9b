"""Pin the oracle (oracle/ref_encoder.py) to golden vectors captured from the ACTUAL reference
(tools/make_goldens.py, run in the build container).  CPU only.  Tolerance: fp32, 2e-5 abs on O(1) tensors
(different but equivalent operation order: closed-form attention, no pad/reshape skew)."""
import os

import numpy as np
import pytest
import torch

from efficientconformer_amd import synth
from efficientconformer_amd.config import build_plan, named_config
from oracle import ref_encoder as R

TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _labels(g):
    offs = g["label_offsets"]
    return [g["labels"][offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]


@pytest.mark.parametrize("tm", [47, 100])
def test_tiny_every_module_output(golden_dir, tm):
    g = _load(golden_dir, "tiny_T%d" % tm)
    cfg = named_config("Tiny")
    plan = build_plan(cfg["encoder_params"])
    sd = synth.make_state_dict(plan, int(g["weight_seed"]), cfg["tokenizer_params"]["vocab_size"])
    mel, lens = synth.make_mel(3, 80, tm, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    assert np.array_equal(lens, g["mel_len"])
    trace = {}
    with torch.no_grad():
        x, out_len = R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(lens), sd, plan, trace)
        logits = R.ctc_logits(x, sd)
    assert np.array_equal(out_len.numpy(), g["out_len"])
    keys = [k[len("trace/"):] for k in g.files if k.startswith("trace/")]
    assert len(keys) == 2 + 5 * len(plan.blocks)
    for k in keys:
        ref = g["trace/" + k]
        got = trace[k].numpy()
        assert got.shape == ref.shape, k
        assert np.abs(got - ref).max() < TOL, (k, np.abs(got - ref).max())
    assert np.abs(x.numpy() - g["out"]).max() < TOL
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL
    assert R.ctc_greedy(logits, out_len) == _labels(g)


def test_small_encoder_and_greedy_labels(golden_dir):
    g = _load(golden_dir, "small_B4_T1001")
    cfg = named_config("EfficientConformerCTCSmall")
    plan = build_plan(cfg["encoder_params"])
    sd = synth.make_state_dict(plan, int(g["weight_seed"]), 256)
    mel, lens = synth.make_mel(4, 80, 1001, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    with torch.no_grad():
        x, out_len = R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(lens), sd, plan)
        logits = R.ctc_logits(x, sd)
    assert out_len.tolist() == g["out_len"].tolist() == [126, 113, 100, 88]
    assert np.abs(x.numpy() - g["out"]).max() < TOL
    assert np.abs(logits[:, ::8].numpy() - g["logits_sample"]).max() < TOL
    # labels: identical wherever the reference's own top-2 margin exceeds the fp32 tolerance
    am = logits.argmax(-1).numpy()
    safe = g["margin"] > 4 * TOL
    assert np.array_equal(am[safe], g["argmax"][safe])
    assert R.ctc_greedy(logits, out_len) == _labels(g)


@pytest.mark.parametrize("name,tm", [("EfficientConformerCTCMedium", 1001), ("EfficientConformerCTCLarge", 1001),
                                     ("EfficientConformerTransducerMedium", 1001), ("ConformerCTCLarge", 501)])
def test_other_configs_sampled(golden_dir, name, tm):
    g = _load(golden_dir, name + "_B2")
    cfg = named_config(name)
    plan = build_plan(cfg["encoder_params"])
    sd = synth.make_state_dict(plan, int(g["weight_seed"]), cfg["tokenizer_params"]["vocab_size"])
    mel, lens = synth.make_mel(2, 80, tm, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    with torch.no_grad():
        x, out_len = R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(lens), sd, plan)
        logits = R.ctc_logits(x, sd)
    assert out_len.tolist() == g["out_len"].tolist()
    assert np.abs(x[:, ::8].numpy() - g["out_rows"]).max() < 5e-5
    assert abs(float(x.double().abs().sum()) - float(g["out_abssum"])) / float(g["out_abssum"]) < 1e-5
    am = logits.argmax(-1).numpy()
    safe = g["margin"] > 2e-4
    assert np.array_equal(am[safe], g["argmax"][safe])


@pytest.mark.parametrize("name", ["att_G1_T47", "att_G3_T47", "att_G3_T48", "att_G1_T126", "att_G3_T250"])
def test_attention_closed_form(golden_dir, name):
    """SURVEY.md section 8a-6 closed form vs the reference's own attention classes (incl. T % G != 0, ragged lens)."""
    g = _load(golden_dir, name)
    sd = {"p.mhsa." + k[2:]: g[k] for k in g.files if k.startswith("w/")}
    with torch.no_grad():
        o, p = R.relpos_attention(torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]), sd, "p",
                                  int(g["heads"]), int(g["group"]), return_probs=True)
    assert np.abs(o.numpy() - g["out"]).max() < TOL
    assert np.abs(p.numpy() - g["probs"]).max() < TOL


def test_mel_frontend_against_independent_dft():
    """a1 is parity-unpinned at the torchaudio boundary; cross-check the restatement against a numpy DFT
    built from the definition (frame, reflect-pad, Hann(400) centred in 512, |rfft|^2, HTK fbank, log)."""
    rng = np.random.Generator(np.random.PCG64(5))
    lens = np.array([4000, 2560], dtype=np.int64)
    audio = synth.make_audio(lens, seed=9)
    mel, mel_len = R.mel_frontend(torch.from_numpy(audio), torch.from_numpy(lens))
    assert mel_len.tolist() == [26, 17] and mel.shape == (2, 80, 26)
    win = np.zeros(512)
    n = np.arange(400)
    win[56:456] = 0.5 - 0.5 * np.cos(2 * np.pi * n / 400)
    fb = R.mel_filterbank().double().numpy()
    for b in range(2):
        xp = np.pad(audio[b].astype(np.float64), (256, 256), mode="reflect")
        for t in (0, 1, 7, 16, 25):
            fr = xp[t * 160:t * 160 + 512] * win
            pw = np.abs(np.fft.rfft(fr)) ** 2
            ref = np.log(pw @ fb + 1e-9)
            assert np.abs(mel[b, :, t].numpy() - ref).max() < 2e-4
    # zero-padded tail frames: exactly log(1e-9) once the 400-tap window has left the audio
    short, _ = R.mel_frontend(torch.from_numpy(audio), torch.from_numpy(lens))
    assert np.allclose(short[1, :, 19:24].numpy(), np.log(np.float32(1e-9)), atol=1e-6)


@pytest.mark.parametrize("name", ["TinyTransducer", "EfficientConformerTransducerMedium"])
def test_rnnt_greedy_oracle_matches_reference_tokens(golden_dir, name):
    """oracle/ref_transducer.greedy_decode vs the reference's own Transducer.gready_search_decoding
    (tools/make_goldens.py: reference_rnnt_greedy) on the reference encoder's outputs: identical token lists,
    both for purely random joint weights (max_consec_dec_step fires on every frame) and with a boosted blank."""
    from efficientconformer_amd.config import named_config
    from oracle import ref_transducer as RT
    g = np.load(os.path.join(golden_dir, "rnnt_%s.npz" % name))
    cfg = named_config(name)
    f, f_len = torch.from_numpy(g["f"]), g["f_len"]
    if name != "TinyTransducer":          # keep the CPU suite short: first utterance's first 40 frames decide the same way
        f_len = np.minimum(f_len, 40)
    for tag in ("rand", "blank"):
        sd = synth.make_transducer_state_dict(f.shape[-1], cfg["decoder_params"], cfg["joint_params"], int(g["weight_seed"]),
                                              blank_bias=float(g["blank_bias_" + tag]))
        toks = RT.greedy_decode(sd, f, f_len, 5)
        offs = g["offsets_" + tag]
        for b, t in enumerate(toks):
            want = g["tokens_" + tag][offs[b]:offs[b + 1]].tolist()
            if name == "TinyTransducer":
                assert t == want, (tag, b)
            else:                          # greedy decoding is causal in the frame index: a prefix of frames gives a prefix of tokens
                assert t == want[:len(t)], (tag, b)
                assert tag != "rand" or len(t) == 5 * int(f_len[b]), (tag, b)


_GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("gname", sorted(f for f in os.listdir(_GOLDEN_DIR) if f.startswith("stream_")))
def test_oracle_streaming_and_causal_vs_reference(gname):
    """Streaming contexts / causal attention + causal depthwise padding (reference encoders.py:68, 94; attentions.py:1377-1403, 506, 1243-1247;
    layers.py:97-101): the oracle against the reference encoder run with `causal` / finite `left_context` / `right_context`."""
    g = np.load(os.path.join(_GOLDEN_DIR, gname))
    small = "small" in gname
    cfg = named_config("EfficientConformerCTCSmall" if small else "Tiny")
    ep = dict(cfg["encoder_params"], **{k[4:]: (bool(g[k]) if k == "cfg/causal" else int(g[k])) for k in g.files if k.startswith("cfg/")})
    plan = build_plan(ep)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(plan, int(g["weight_seed"]), cfg["tokenizer_params"]["vocab_size"]).items()}
    lens = g["mel_len"].tolist()
    mel, ln = synth.make_mel(len(lens), 80, max(lens), lens, seed=int(g["mel_seed"]))
    with torch.no_grad():
        out, out_len = R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(ln), sd, plan)
    assert out_len.tolist() == g["out_len"].tolist()
    ref = torch.from_numpy(g["out_rows"] if small else g["out"])
    got = out[:, ::4] if small else out
    assert float((got - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("gname", ["tiny_T47.npz", "tiny_T100.npz"])
def test_oracle_attention_maps_equal_the_reference(golden_dir, gname):
    """The third return value of ConformerEncoder.forward (encoders.py:126-142: one (B, H, Tg, Tg) softmax map per block): the oracle's maps
    of the first and the last block against the reference's (tools/make_goldens.py stores atts[0] / atts[-1])."""
    g = np.load(os.path.join(golden_dir, gname))
    plan = build_plan(named_config("Tiny")["encoder_params"])
    sd = synth.make_state_dict(plan, int(g["weight_seed"]), None)
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    b, tm = len(g["mel_len"]), int(g["mel_len"].max())
    mel, ln = synth.make_mel(b, 80, tm, g["mel_len"].tolist(), seed=int(g["mel_seed"]))
    trace = {}
    with torch.no_grad():
        R.encoder_from_mel(torch.from_numpy(mel), torch.from_numpy(ln), sd, plan, trace)
    first, last = trace["blocks.0.att_w"], trace["blocks.%d.att_w" % (len(plan.blocks) - 1)]
    assert tuple(first.shape) == g["att0"].shape and tuple(last.shape) == g["att_last"].shape
    assert float((first - torch.from_numpy(g["att0"])).abs().max()) < 2e-6
    assert float((last - torch.from_numpy(g["att_last"])).abs().max()) < 2e-6
