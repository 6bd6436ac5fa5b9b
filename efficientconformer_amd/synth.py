"""Deterministic synthetic weights and inputs (no checkpoints or datasets are reachable offline).

Weights: a *key-seeded* generator — every state-dict key gets its own
``numpy`` PCG64 stream seeded with ``crc32(key) ^ seed`` — so the same tensors
are produced in the build container (where the golden vectors are captured from
the reference, tools/make_goldens.py) and on the GPU box.  numpy's PCG64 +
``standard_normal`` is specified to be reproducible across platforms, unlike
``torch.randn``.

Inputs follow SURVEY.md section 8(d): mel ~ N(-5.65, 4.23^2) (the dataset statistics
recorded in the reference configs, configs/EfficientConformerCTCSmall.json:36-37),
audio = 0.1*randn clipped to [-1, 1] and zeroed beyond each length (collate
zero-pad, reference utils/preprocessing.py:38), lengths sorted descending
(utils/preprocessing.py:33).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List

import numpy as np

from . import params as P
from .config import EncoderPlan


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) ^ seed) & 0xFFFFFFFF))


def make_tensor(key: str, shape, kind: str, seed: int) -> np.ndarray:
    g = _rng(key, seed)
    if kind == P.W:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        return (g.standard_normal(shape) / math.sqrt(fan_in)).astype(np.float32)
    if kind == P.B:
        return (0.02 * g.standard_normal(shape)).astype(np.float32)
    if kind == P.GAMMA:
        return (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
    if kind == P.BETA:
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if kind == P.RMEAN:
        return (0.2 * g.standard_normal(shape)).astype(np.float32)
    if kind == P.RVAR:
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if kind == P.NBT:
        return np.zeros(shape, dtype=np.int64)
    if kind == P.UV:
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if kind == P.EMB:                      # nn.Embedding init N(0, 1) with the padding_idx = 0 row zeroed (decoders.py:46)
        e = g.standard_normal(shape).astype(np.float32)
        e[0] = 0.0
        return e
    raise ValueError(kind)


def make_state_dict(plan: EncoderPlan, seed: int = 0, vocab: int | None = None, prefix: str = "") -> Dict[str, np.ndarray]:
    """Key-seeded weights for the encoder (keys without prefix unless given) and optional ``fc`` head."""
    sd = {}
    for key, shape, kind in P.param_specs(plan):
        sd[prefix + key] = make_tensor(key, shape, kind, seed)
    if vocab is not None:
        for key, shape, kind in P.head_specs(plan, vocab):
            sd[key] = make_tensor(key, shape, kind, seed)
    return sd


def make_transducer_state_dict(dim_encoder: int, decoder_params: dict, joint_params: dict, seed: int = 0,
                               blank_bias: float = 0.0) -> Dict[str, np.ndarray]:
    """Key-seeded prediction / joint network weights (keys ``decoder.*`` / ``joint_network.*``).

    ``blank_bias`` is added to the blank logit's bias: with purely random weights the blank wins 1/V of the decisions and
    greedy decoding emits ``max_consec_dec_step`` tokens on every frame (a useful stress case, but not what a trained
    model does); ~1.2 gives the roughly 1 token per 3-4 encoder frames of a trained LibriSpeech model."""
    sd = {}
    for key, shape, kind in P.transducer_specs(dim_encoder, decoder_params, joint_params):
        sd[key] = make_tensor(key, shape, kind, seed)
    sd["joint_network.linear_joint.bias"][0] += np.float32(blank_bias)
    return sd


def make_mel(batch: int, n_mels: int, tm: int, lengths: List[int] | None = None, seed: int = 4321):
    """Seeded mel batch (B, n_mels, Tm) float32 + int64 mel lengths (sorted descending)."""
    g = np.random.Generator(np.random.PCG64(seed))
    mel = (-5.6501 + 4.2280 * g.standard_normal((batch, n_mels, tm))).astype(np.float32)
    if lengths is None:
        lengths = [tm] * batch
    lens = np.asarray(sorted(lengths, reverse=True), dtype=np.int64)
    assert lens.max() <= tm
    return mel, lens


def libri_lengths(batch: int, seed: int = 1234, sample_rate: int = 16000) -> np.ndarray:
    """'LibriSpeech-shaped' utterance lengths in samples (SURVEY.md section 8d W-libri):
    16000 * clip(lognormal(ln 11, 0.45), 1.5, 16.0) s, sorted descending within the batch."""
    g = np.random.Generator(np.random.PCG64(seed))
    sec = np.clip(g.lognormal(math.log(11.0), 0.45, batch), 1.5, 16.0)
    return np.sort((sec * sample_rate).astype(np.int64))[::-1].copy()


def make_audio(lengths, seed: int = 1234) -> np.ndarray:
    """Audio (B, L_max) float32: 0.1*randn clipped to [-1,1], rows zeroed beyond each length."""
    lengths = np.asarray(lengths, dtype=np.int64)
    g = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    x = np.clip(0.1 * g.standard_normal((len(lengths), int(lengths.max())), dtype=np.float32), -1.0, 1.0)
    for b, n in enumerate(lengths):
        x[b, n:] = 0.0
    return x
