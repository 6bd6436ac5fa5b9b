#!/usr/bin/env python3
"""Capture golden vectors by executing the ACTUAL reference (build container only).

Imports burchim/EfficientConformer from /root/reference with import-time stubs for
its absent third-party dependencies, loads key-seeded
weights (efficientconformer_amd/synth.py), drives the reference encoder *from
mel* on seeded inputs and writes small fixtures to tests/golden/.  Only tensors
leave this script: no reference source, bytecode or text is written anywhere.
It refuses to run where /root/reference is absent (e.g. the GPU box).

    python tools/make_goldens.py            # regenerate every fixture
"""
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from efficientconformer_amd import synth  # noqa: E402
from efficientconformer_amd.config import build_plan, named_config  # noqa: E402


def import_reference():
    if not os.path.isdir(REF):
        sys.exit("make_goldens: %s not present; goldens can only be regenerated in the build container" % REF)

    class _Any(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x, *a, **k):
            return x
    ta = types.ModuleType("torchaudio")
    tr = types.ModuleType("torchaudio.transforms")
    for n in ("Spectrogram", "MelScale", "FrequencyMasking", "TimeMasking"):
        setattr(tr, n, _Any)
    ta.transforms = tr
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tr
    sys.path.insert(0, REF)
    import models.attentions as att
    import models.encoders as enc
    return enc, att


def reference_rnnt_greedy(f, f_len, dec_params, joint_params, sd):
    """Execute the reference's own Transducer.gready_search_decoding (transducer.py:139-186) on given encoder
    outputs: a Transducer object is assembled around the reference's RnnDecoder / JointNetwork classes (without
    Model.__init__: no tokenizer file, optimizer or loss here) and its real method is called."""
    _stub_third_party()
    import models.transducer as tr
    import models.decoders as dec
    import models.joint_networks as jn
    obj = tr.Transducer.__new__(tr.Transducer)
    nn.Module.__init__(obj)
    obj.decoder = dec.RnnDecoder(dec_params).eval()
    obj.joint_network = jn.JointNetwork(f.shape[-1], dec_params["dim_model"], dec_params["vocab_size"], joint_params).eval()
    obj.decoder.load_state_dict(to_torch({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}), strict=True)
    obj.joint_network.load_state_dict(to_torch({k[len("joint_network."):]: v for k, v in sd.items() if k.startswith("joint_network.")}), strict=True)
    obj.max_consec_dec_step = dec_params.get("max_consec_dec_step", 5)

    class _Ids:                                   # tokenizer.decode(list of id lists) -> keep the ids
        @staticmethod
        def decode(lists):
            return [list(map(int, l)) for l in lists]
    obj.tokenizer = _Ids()
    object.__setattr__(obj, "encoder", lambda x, x_len: (f, f_len, None))
    obj.eval()
    with torch.no_grad():
        return obj.gready_search_decoding(torch.zeros(f.shape[0], 1), f_len)


def to_torch(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def run_encoder(enc_mod, name, mel, lens, seed, hooks=False, extra=None):
    """Reference ConformerEncoder driven from mel (encoders.py:107-140) + fc head.  extra: encoder_params overrides (causal / contexts)."""
    cfg = named_config(name)
    if extra:
        cfg["encoder_params"] = dict(cfg["encoder_params"], **extra)
    plan = build_plan(cfg["encoder_params"])
    vocab = cfg["tokenizer_params"]["vocab_size"]
    sd = synth.make_state_dict(plan, seed, vocab)
    model = enc_mod.ConformerEncoder(cfg["encoder_params"]).eval()
    missing = model.load_state_dict(to_torch({k: v for k, v in sd.items() if not k.startswith("fc.")}), strict=True)
    fc = nn.Linear(plan.dim_out, vocab).eval()
    fc.load_state_dict({"weight": torch.from_numpy(sd["fc.weight"]), "bias": torch.from_numpy(sd["fc.bias"])})
    trace = {}
    handles = []
    if hooks:
        def grab(key):
            def fn(mod, inp, out):
                trace[key] = (out[0] if isinstance(out, tuple) else out).detach().clone()
            return fn
        handles.append(model.subsampling_module.register_forward_hook(grab("subsample")))
        handles.append(model.linear.register_forward_hook(grab("linear")))
        for i, blk in enumerate(model.blocks):
            p = "blocks.%d" % i
            handles.append(blk.feed_forward_module1.register_forward_hook(grab(p + ".ffn1")))
            handles.append(blk.multi_head_self_attention_module.register_forward_hook(grab(p + ".mhsa")))
            handles.append(blk.convolution_module.register_forward_hook(grab(p + ".conv")))
            handles.append(blk.feed_forward_module2.register_forward_hook(grab(p + ".ffn2")))
            handles.append(blk.register_forward_hook(grab(p + ".out")))
    model.preprocessing.forward = lambda x, l: (x, l)      # frontend lives in torchaudio (absent): start from mel
    with torch.no_grad():
        x, out_len, atts = model(torch.from_numpy(mel), torch.from_numpy(lens))
        logits = fc(x)
    for h in handles:
        h.remove()
    return plan, x, out_len, logits, atts, trace


def _stub_third_party():
    for n, attrs in (("jiwer", ()), ("kenlm", ()), ("warp_rnnt", ("rnnt_loss",)), ("ctcdecode", ("CTCBeamDecoder",)),
                     ("torch.utils.tensorboard", ("SummaryWriter",)), ("sentencepiece", ("SentencePieceProcessor", "SentencePieceTrainer"))):
        if n not in sys.modules or n == "sentencepiece":
            m = types.ModuleType(n)
            for a in attrs:
                setattr(m, a, object)
            sys.modules.setdefault(n, m)


def greedy_reference(logits, lens):
    """The reference's OWN greedy method (models/model_ctc.py:90-136), called as imported: a stand-in `self` whose encoder returns the
    given logits, whose fc is the identity and whose tokenizer keeps the id lists."""
    _stub_third_party()
    import models.model_ctc as mc
    fake = types.SimpleNamespace(encoder=lambda x, x_len: (x, x_len, None), fc=lambda x: x,
                                 tokenizer=types.SimpleNamespace(decode=lambda lists: [list(map(int, l)) for l in lists]))
    with torch.no_grad():
        return mc.ModelCTC.gready_search_decoding(fake, logits, lens)


def pack_labels(lists):
    flat = np.asarray([t for l in lists for t in l], dtype=np.int32)
    offs = np.cumsum([0] + [len(l) for l in lists]).astype(np.int32)
    return flat, offs


def margins(logits):
    top2 = logits.topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def rnnt_goldens(enc_mod):
    """RNN-T greedy decode (BASELINE.json configs[3]): the reference encoder's own output f -> the reference's greedy
    token lists, for purely random joint weights (blank almost never wins: the max_consec_dec_step rule fires on every
    frame) and with a boosted blank bias (trained-model-like token rate)."""
    for name, tm, lens, seed in (("TinyTransducer", 100, [100, 77, 52, 9], 7), ("EfficientConformerTransducerMedium", 1001, [1001, 640], 0)):
        cfg = named_config(name)
        mel, ln = synth.make_mel(len(lens), 80, tm, lens, seed=4321)
        plan, f, f_len, _, _, _ = run_encoder(enc_mod, name, mel, ln, seed=seed)
        arrs = {"mel_seed": np.int64(4321), "weight_seed": np.int64(seed), "mel_len": ln, "f": f.numpy(), "f_len": f_len.numpy()}
        for tag, bb in (("rand", 0.0), ("blank", 1.2)):
            sd = synth.make_transducer_state_dict(plan.dim_out, cfg["decoder_params"], cfg["joint_params"], seed, blank_bias=bb)
            toks = reference_rnnt_greedy(f, f_len, cfg["decoder_params"], cfg["joint_params"], sd)
            flat, offs = pack_labels(toks)
            arrs["tokens_" + tag], arrs["offsets_" + tag], arrs["blank_bias_" + tag] = flat, offs, np.float32(bb)
            print("  %s/%s: tokens per utterance %s (frames %s)" % (name, tag, [len(t) for t in toks], f_len.tolist()))
        save("rnnt_" + name, **arrs)


STREAMING = (("causal", dict(causal=True)), ("ctx_l20_r4", dict(left_context=20, right_context=4)),
             ("causal_l12", dict(causal=True, left_context=12)), ("ctx_l3_r0", dict(left_context=3, right_context=0)))


def streaming_goldens(enc_mod):
    """Streaming / causal contexts (reference encoders.py:68, 94; attentions.py:1377-1403, 506; layers.py:94-101): the reference encoder
    with `causal` and finite `left_context` / `right_context`, tiny config (both sequence lengths) and EfficientConformerCTCSmall."""
    for tag, extra in STREAMING:
        for tm, lens in ((47, [47, 40, 23]), (100, [100, 77, 52])):
            mel, ln = synth.make_mel(3, 80, tm, lens, seed=4321 + tm)
            plan, x, out_len, logits, atts, _ = run_encoder(enc_mod, "Tiny", mel, ln, seed=7, extra=extra)
            lab, offs = pack_labels(greedy_reference(logits, out_len))
            save("stream_tiny_%s_T%d" % (tag, tm), mel_seed=np.int64(4321 + tm), weight_seed=np.int64(7), mel_len=ln, out=x.numpy(),
                 out_len=out_len.numpy(), labels=lab, label_offsets=offs, **{"cfg/" + k: np.int64(v) for k, v in extra.items()})
    for tag, extra in (("causal", dict(causal=True)), ("ctx_l64_r16", dict(left_context=64, right_context=16))):
        mel, ln = synth.make_mel(2, 80, 601, [601, 433], seed=4321)
        plan, x, out_len, logits, atts, _ = run_encoder(enc_mod, "EfficientConformerCTCSmall", mel, ln, seed=0, extra=extra)
        save("stream_small_%s" % tag, mel_seed=np.int64(4321), weight_seed=np.int64(0), mel_len=ln, out_rows=x[:, ::4].numpy(),
             out_len=out_len.numpy(), argmax=logits.argmax(-1).numpy().astype(np.int16), margin=margins(logits),
             **{"cfg/" + k: np.int64(v) for k, v in extra.items()})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    enc_mod, att_mod = import_reference()
    if "--only-rnnt" in sys.argv:
        return rnnt_goldens(enc_mod)
    if "--only-streaming" in sys.argv:
        return streaming_goldens(enc_mod)
    rnnt_goldens(enc_mod)
    streaming_goldens(enc_mod)

    # ---- 1. tiny config: every module output, two sequence lengths (T1 % 3 == 0 and != 0)
    for tm, lens in ((47, [47, 40, 23]), (100, [100, 77, 52])):
        mel, ln = synth.make_mel(3, 80, tm, lens, seed=4321 + tm)
        plan, x, out_len, logits, atts, trace = run_encoder(enc_mod, "Tiny", mel, ln, seed=7, hooks=True)
        arrs = {"mel_seed": np.int64(4321 + tm), "weight_seed": np.int64(7), "mel_len": ln,
                "out": x.numpy(), "out_len": out_len.numpy(), "logits": logits.numpy(),
                "att0": atts[0].numpy(), "att_last": atts[-1].numpy()}
        for k, v in trace.items():
            arrs["trace/" + k] = v.numpy()
        lab, offs = pack_labels(greedy_reference(logits, out_len))
        arrs["labels"], arrs["label_offsets"] = lab, offs
        save("tiny_T%d" % tm, **arrs)

    # ---- 2. EfficientConformerCTCSmall, B=4, Tm=1001, ragged
    mel, ln = synth.make_mel(4, 80, 1001, [1001, 900, 800, 700], seed=4321)
    plan, x, out_len, logits, atts, _ = run_encoder(enc_mod, "EfficientConformerCTCSmall", mel, ln, seed=0)
    lab, offs = pack_labels(greedy_reference(logits, out_len))
    save("small_B4_T1001", mel_seed=np.int64(4321), weight_seed=np.int64(0), mel_len=ln,
         out=x.numpy(), out_len=out_len.numpy(), argmax=logits.argmax(-1).numpy().astype(np.int16),
         margin=margins(logits), labels=lab, label_offsets=offs,
         logits_sample=logits[:, ::8].numpy())

    # ---- 3. other BASELINE.json configs: sampled rows + labels + margins (B=2, ragged)
    for name, tm, lens in (("EfficientConformerCTCMedium", 1001, [1001, 640]),
                           ("EfficientConformerCTCLarge", 1001, [1001, 640]),
                           ("EfficientConformerTransducerMedium", 1001, [1001, 640]),
                           ("ConformerCTCLarge", 501, [501, 333])):
        mel, ln = synth.make_mel(2, 80, tm, lens, seed=4321)
        plan, x, out_len, logits, atts, _ = run_encoder(enc_mod, name, mel, ln, seed=0)
        lab, offs = pack_labels(greedy_reference(logits, out_len))
        save(name + "_B2", mel_seed=np.int64(4321), weight_seed=np.int64(0), mel_len=ln,
             out_rows=x[:, ::8].numpy(), out_sum=np.float64(x.double().sum().item()),
             out_abssum=np.float64(x.double().abs().sum().item()), out_len=out_len.numpy(),
             argmax=logits.argmax(-1).numpy().astype(np.int16), margin=margins(logits),
             labels=lab, label_offsets=offs)

    # ---- 4. op-level: the reference's attention classes on their own (closed-form pin)
    for group, t, lens in ((1, 47, [47, 30]), (3, 47, [47, 31]), (3, 48, [48, 9]), (1, 126, [126, 88]), (3, 250, [250, 101])):
        dim, heads = 48, 4
        g = np.random.Generator(np.random.PCG64(1000 + 10 * t + group))
        xin = g.standard_normal((2, t, dim)).astype(np.float32)
        keys = [("u", (dim,), "uv"), ("v", (dim,), "uv")]
        for n in ("query_layer", "key_layer", "value_layer", "output_layer", "pos_layer"):
            keys += [(n + ".weight", (dim, dim), "weight"), (n + ".bias", (dim,), "bias")]
        sd = {k: synth.make_tensor("att." + k, s, kind, 3) for k, s, kind in keys}
        if group > 1:
            mod = att_mod.GroupedRelPosMultiHeadSelfAttention(dim, heads, False, 600, group).eval()
        else:
            mod = att_mod.RelPosMultiHeadSelfAttention(dim, heads, False, 600).eval()
        mod.load_state_dict(to_torch(sd), strict=True)
        ln = torch.tensor(lens)
        mask = att_mod.StreamingMask(600, 600)(torch.zeros(2, 1, t), ln)
        with torch.no_grad():
            xt = torch.from_numpy(xin)
            o, w, _ = mod(xt, xt, xt, mask)
        arrs = {"x": xin, "lens": np.asarray(lens, dtype=np.int64), "out": o.numpy(), "probs": w.numpy(),
                "group": np.int64(group), "heads": np.int64(heads)}
        arrs.update({"w/" + k: v for k, v in sd.items()})
        save("att_G%d_T%d" % (group, t), **arrs)


if __name__ == "__main__":
    main()
