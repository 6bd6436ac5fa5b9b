"""Encoder configuration: the reference's ``encoder_params`` dict -> per-block plan.

The reference resolves per-block hyper-parameters from per-stage lists inside
``ConformerEncoder.__init__`` (reference models/encoders.py:80-95).  This module
restates that *rule* (not the code) once, so that the Python host, the C-ABI
(`EcConfig` in include/effconf.h) and the oracle all agree on the block plan:

    for block b:  n_gt(L) = #{e in L : b >  e},  n_ge(L) = #{e in L : b >= e}
      dim_model, num_heads         <- stage n_gt(expand_blocks)
      dim_expand, kernel_size      <- stage n_ge(expand_blocks)
      att_group_size               <- stage n_gt(strided_blocks)
      max_pos_encoding             <- max_pos // stride**n_gt(strided_blocks)
      conv_stride                  <- conv_stride[stage] if b in strided_blocks else 1
    streaming / causal (encoders.py:68, 94; attentions.py:1377-1403, 506; layers.py:94-101; blocks.py:83):
      left_context  L (default max_pos_encoding), right_context R (default max_pos_encoding; 0 when causal), in frames AFTER the
      subsampling: key j of query i is masked iff j - i > R or j - i < -L; the mask is sliced `::s` after every strided block, so
      block b compares mask_stride * (j - i), mask_stride = product of the strides of the blocks before b (and `::G` inside grouped
      attention: att_group_size * mask_stride * (j' - i') for grouped positions);
      causal: relative table of T (not 2T - 1) positions, depthwise conv pre-padded (k - 1, 0) instead of ((k-1)/2, (k-1)/2)

Only the values shipped configs reach are implemented natively; anything else
raises NotImplementedError (reference raises a bare Exception for unknown
sub-modules, encoders.py:65).
"""
from __future__ import annotations

import copy
import json
from dataclasses import dataclass, field
from typing import List, Optional


def _stage(value, idx, default=None):
    if value is None:
        return default
    if isinstance(value, (list, tuple)):
        return value[idx]
    return value


@dataclass
class BlockPlan:
    index: int
    dim_model: int
    dim_expand: int
    dim_ffn1: int
    dim_ffn2: int
    num_heads: int
    kernel_size: int
    group_size: int
    max_pos: int
    conv_stride: int
    mask_stride: int = 1      # product of the conv strides of the blocks before this one (the streaming mask is sliced ::s after each)

    @property
    def dim_head(self) -> int:
        """Grouped head dim d = G*D/H (reference attentions.py:643)."""
        return self.group_size * self.dim_model // self.num_heads

    @property
    def transition(self) -> bool:
        return self.dim_model != self.dim_expand


@dataclass
class EncoderPlan:
    n_mels: int
    sample_rate: int
    n_fft: int
    win_length: int
    hop_length: int
    normalize: bool
    mean: float
    std: float
    sub_layers: int
    sub_filters: List[int]
    dim_in: int               # subsampling_filters[-1] * n_mels // 2**layers
    causal: bool = False      # encoders.py:94: causal rel-pos table + causal depthwise padding + right_context 0
    left_context: int = 1 << 30    # frames after the subsampling; >= any sequence length = unlimited (the shipped configs: max_pos_encoding)
    right_context: int = 1 << 30
    blocks: List[BlockPlan] = field(default_factory=list)

    @property
    def dim_out(self) -> int:
        return self.blocks[-1].dim_expand

    def lengths(self, n_samples: int):
        """(Tm, T1 after subsampling, [T after each block]) for an audio length in samples."""
        tm = n_samples // self.hop_length + 1            # modules.py:100
        return (tm,) + self.lengths_from_mel(tm)

    def lengths_from_mel(self, tm: int):
        t = tm
        for _ in range(self.sub_layers):                  # modules.py:243
            t = (t - 1) // 2 + 1
        t1 = t
        ts = []
        for b in self.blocks:                             # encoders.py:139
            if b.conv_stride > 1:
                t = (t - 1) // b.conv_stride + 1
            ts.append(t)
        return t1, ts


def build_plan(params: dict) -> EncoderPlan:
    """Validate an ``encoder_params`` dict and resolve the per-block plan."""
    p = params
    if p.get("arch", "Conformer") != "Conformer":
        raise Exception("Unknown encoder architecture:", p.get("arch"))
    if p["subsampling_module"] != "Conv2d":
        if p["subsampling_module"] in ("Conv1d", "Conv2dPool", "VGG"):
            raise NotImplementedError(
                "subsampling_module=%r is not used by any shipped config; only Conv2d is native"
                % p["subsampling_module"])
        raise Exception("Unknown subsampling module:", p["subsampling_module"])
    if not p.get("relative_pos_enc", False):
        raise NotImplementedError("absolute positional encodings (relative_pos_enc=false) not native")
    if p.get("linear_att", False) or p.get("att_kernel_size", None) is not None:
        raise NotImplementedError("linear / local attention variants not native")
    if p.get("subsampling_norm", "batch") != "batch" or p.get("subsampling_act", "swish") != "swish":
        raise NotImplementedError("subsampling norm/act other than batch/swish not native")
    if int(p.get("subsampling_kernel_size", 3)) != 3:
        raise NotImplementedError("subsampling kernel size != 3 not native")

    expand = list(p.get("expand_blocks", []) or [])
    strided = list(p.get("strided_blocks", []) or [])
    stride = int(p.get("stride", 2))
    hop = int(p["sample_rate"] * p["hop_length_ms"]) // 1000
    win = int(p["sample_rate"] * p["win_length_ms"]) // 1000
    layers = int(p["subsampling_layers"])
    filters = list(p["subsampling_filters"])
    plan = EncoderPlan(
        n_mels=int(p["n_mels"]), sample_rate=int(p["sample_rate"]), n_fft=int(p["n_fft"]),
        win_length=win, hop_length=hop, normalize=bool(p.get("normalize", False)),
        mean=float(p.get("mean", 0.0)), std=float(p.get("std", 1.0)),
        sub_layers=layers, sub_filters=filters,
        dim_in=filters[-1] * int(p["n_mels"]) // 2 ** layers)
    # StreamingMask(left_context, right_context) of encoders.py:68
    plan.causal = bool(p.get("causal", False))
    plan.left_context = int(p.get("left_context", p["max_pos_encoding"]))
    plan.right_context = 0 if plan.causal else int(p.get("right_context", p["max_pos_encoding"]))
    if plan.left_context < 0 or plan.right_context < 0:
        raise Exception("left_context / right_context must be >= 0")
    mask_stride = 1
    for b in range(int(p["num_blocks"])):
        n_gt_e = sum(1 for e in expand if b > e)
        n_ge_e = sum(1 for e in expand if b >= e)
        n_gt_s = sum(1 for s in strided if b > s)
        att_stride = _stage(p.get("att_stride", 1), n_gt_s, 1) if b in strided else 1
        if att_stride != 1:
            raise NotImplementedError("att_stride > 1 not native (no shipped config uses it)")
        cs = _stage(p.get("conv_stride", 1), n_gt_s, 1) if b in strided else 1
        d_model = int(_stage(p["dim_model"], n_gt_e))
        d_exp = int(_stage(p["dim_model"], n_ge_e))
        g = int(_stage(p.get("att_group_size", 1), n_gt_s, 1))
        heads = int(_stage(p["num_heads"], n_gt_e))
        if g > 1 and g % 2 == 0:
            raise NotImplementedError("even att_group_size (reference table duplicates position 0)")
        if (g * d_model) % heads != 0:
            raise Exception("dim_model * group_size must be divisible by num_heads")
        if d_model % 4 or d_exp % 4:
            raise NotImplementedError("model dims must be multiples of 4 (16-byte fp32 rows) for the HIP path")
        plan.blocks.append(BlockPlan(
            index=b, dim_model=d_model, dim_expand=d_exp,
            dim_ffn1=d_model * int(p["ff_ratio"]), dim_ffn2=d_exp * int(p["ff_ratio"]),
            num_heads=heads, kernel_size=int(_stage(p["kernel_size"], n_ge_e)),
            group_size=g, max_pos=int(p["max_pos_encoding"]) // stride ** n_gt_s,
            conv_stride=int(cs), mask_stride=mask_stride))
        mask_stride *= int(cs)
    return plan


# --------------------------------------------------------------------------
# Named model configurations (the shipped JSON files of the reference,
# reference configs/*.json, expressed programmatically; a JSON path or dict in
# the reference's format is accepted everywhere a name is).
# --------------------------------------------------------------------------

_FRONTEND = dict(sample_rate=16000, win_length_ms=25, hop_length_ms=10, n_fft=512, n_mels=80,
                 normalize=False, mean=-5.6501, std=4.2280,
                 spec_augment=True, mF=2, F=27, mT=5, pS=0.05)


def _eff(dims, heads, blocks, se, filt):
    return dict(arch="Conformer", num_blocks=blocks, dim_model=list(dims), ff_ratio=4, num_heads=heads,
                kernel_size=15, Pdrop=0.1, conv_stride=2, att_stride=1, strided_blocks=list(se),
                expand_blocks=list(se), att_group_size=[3, 1, 1], relative_pos_enc=True,
                max_pos_encoding=10000, subsampling_module="Conv2d", subsampling_layers=1,
                subsampling_filters=[filt], subsampling_kernel_size=3, subsampling_norm="batch",
                subsampling_act="swish", **_FRONTEND)


def _plain(dim, heads, blocks):
    return dict(arch="Conformer", num_blocks=blocks, dim_model=dim, ff_ratio=4, num_heads=heads,
                kernel_size=31, Pdrop=0.1, relative_pos_enc=True, max_pos_encoding=10000,
                subsampling_module="Conv2d", subsampling_layers=2, subsampling_filters=[dim, dim],
                subsampling_kernel_size=3, subsampling_norm="batch", subsampling_act="swish", **_FRONTEND)


_NAMED = {
    "EfficientConformerCTCSmall": ("CTC", _eff([120, 168, 240], 4, 15, [4, 9], 120), 256),
    "EfficientConformerCTCMedium": ("CTC", _eff([180, 256, 360], 4, 16, [4, 10], 180), 256),
    "EfficientConformerCTCLarge": ("CTC", _eff([360, 512, 720], 8, 16, [4, 10], 360), 256),
    "EfficientConformerTransducerSmall": ("Transducer", _eff([100, 140, 200], 4, 15, [4, 9], 100), 1000),
    "EfficientConformerTransducerMedium": ("Transducer", _eff([180, 256, 360], 4, 15, [4, 9], 180), 1000),
    "EfficientConformerTransducerLarge": ("Transducer", _eff([360, 512, 720], 8, 15, [4, 9], 360), 1000),
    "ConformerCTCSmall": ("CTC", _plain(176, 4, 16), 256),
    "ConformerCTCMedium": ("CTC", _plain(256, 4, 18), 256),
    "ConformerCTCLarge": ("CTC", _plain(512, 8, 18), 256),
    "ConformerTransducerSmall": ("Transducer", _plain(144, 6, 16), 1000),
    "ConformerTransducerMedium": ("Transducer", _plain(256, 4, 16), 1000),
    "ConformerTransducerLarge": ("Transducer", _plain(512, 8, 17), 1000),
    # not a shipped model: small dims for unit tests (exercises grouping, T % G != 0, both transitions)
    "Tiny": ("CTC", dict(_eff([24, 32, 48], 4, 6, [1, 3], 24), max_pos_encoding=2000), 32),
    # not a shipped model: the Small topology with widths whose per-head spans are 16-byte aligned (alignment experiments)
    "SmallAligned": ("CTC", _eff([128, 192, 256], 4, 15, [4, 9], 128), 256),
    "TinyTransducer": ("Transducer", dict(_eff([24, 32, 48], 4, 6, [1, 3], 24), max_pos_encoding=2000), 40),
}

# decoder_params["dim_model"] == joint_params["dim_model"] of the shipped Transducer configs
# (reference configs/*Transducer*.json: 640 for Medium / Large, 320 for Small)
_RNNT_DIM = {"EfficientConformerTransducerSmall": 320, "EfficientConformerTransducerMedium": 640, "EfficientConformerTransducerLarge": 640,
             "ConformerTransducerSmall": 320, "ConformerTransducerMedium": 640, "ConformerTransducerLarge": 640, "TinyTransducer": 32}


def named_config(name: str) -> dict:
    """Return a config dict in the reference's JSON layout for a model name."""
    if name not in _NAMED:
        raise KeyError("unknown model %r; known: %s" % (name, sorted(_NAMED)))
    mtype, enc, vocab = _NAMED[name]
    cfg = {"model_name": name, "model_type": mtype, "encoder_params": copy.deepcopy(enc),
           "tokenizer_params": {"vocab_type": "bpe", "vocab_size": vocab}}
    if mtype == "Transducer":
        d = _RNNT_DIM[name]
        cfg["decoder_params"] = {"arch": "RNN", "num_layers": 1, "dim_model": d, "vocab_size": vocab}
        cfg["joint_params"] = {"joint_mode": "sum", "dim_model": d, "act": "tanh"}
    return cfg


def load_config(cfg) -> dict:
    """Accept a model name, a path to a reference-format JSON file, or a dict."""
    if isinstance(cfg, dict):
        return cfg
    if isinstance(cfg, str) and cfg in _NAMED:
        return named_config(cfg)
    with open(cfg) as f:
        return json.load(f)
