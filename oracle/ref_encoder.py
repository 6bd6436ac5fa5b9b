"""ORACLE — test infrastructure only, never part of the product path.

CPU (torch fp32, eager) restatement of the Efficient Conformer encoder forward
path of burchim/EfficientConformer, written from the algorithm (closed forms of
SURVEY.md section 8a), not from the reference's module tree.  Every function cites the
reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``efficientconformer_amd``) never does and fails loudly without its HIP library.

Pinning: functions a2-a10 + CTC head are pinned against golden vectors captured
by executing the *actual reference* in the build container
(tools/make_goldens.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
The mel frontend (a1) is **parity unpinned**: its arithmetic lives in
``torchaudio`` (un-vendored, un-pinned dependency of the reference, absent from
this image; reference call sites models/modules.py:81-82, 90, 93).  It restates
torchaudio's published ``Spectrogram(power=2)`` + ``MelScale(htk, norm=None)``
algorithm and is cross-checked only against an independent numpy DFT.

All tensors are torch CPU float32; ``sd`` is a state dict with the reference's
key names without the ``encoder.`` prefix (efficientconformer_amd/params.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

LN_EPS = 1e-6   # every LayerNorm on the path: modules.py:377, 447, 500; blocks.py:97
BN_EPS = 1e-5   # nn.BatchNorm default: modules.py:227, 505


def _t(sd, key) -> torch.Tensor:
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


# --------------------------------------------------------------------------
# a1  mel frontend  (reference models/modules.py:77-106; torchaudio Spectrogram + MelScale)
# --------------------------------------------------------------------------

def mel_filterbank(n_freqs: int = 257, n_mels: int = 80, f_min: float = 0.0, f_max: float = 8000.0,
                   sample_rate: int = 16000) -> torch.Tensor:
    """HTK triangular filterbank (n_freqs, n_mels), no area normalisation
    (torchaudio.functional.melscale_fbanks(mel_scale='htk', norm=None), as instantiated at modules.py:82)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # (n_freqs, n_mels+2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_frontend(audio: torch.Tensor, x_len: Optional[torch.Tensor], n_fft=512, win_length=400,
                 hop_length=160, n_mels=80, sample_rate=16000, normalize=False, mean=0.0, std=1.0):
    """(B, L) audio -> (B, n_mels, L//hop+1) log-mel; lengths L_b//hop+1 (modules.py:87-106)."""
    window = torch.hann_window(win_length, periodic=True)
    spec = torch.stft(audio.float(), n_fft, hop_length, win_length, window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                       # Spectrogram(power=2)   modules.py:90
    fb = mel_filterbank(n_fft // 2 + 1, n_mels, 0.0, 8000.0, sample_rate)
    mel = torch.matmul(power.transpose(1, 2), fb).transpose(1, 2)  # MelScale               modules.py:93
    mel = (mel + 1e-9).log()                                       # modules.py:96
    if x_len is not None:
        x_len = torch.div(x_len, hop_length, rounding_mode="floor") + 1   # modules.py:100
    if normalize:
        mel = (mel - mean) / std                                   # modules.py:104
    return mel, x_len


# --------------------------------------------------------------------------
# a2  Conv2d subsampling  (modules.py:232-249)
# --------------------------------------------------------------------------

def subsample(mel: torch.Tensor, lens: Optional[torch.Tensor], sd, n_layers: int):
    """(B, F, T) -> (B, C*F/2^L, T/2^L): [conv3x3 s2 p1 -> BN(eval) -> Swish] x L, channel-major flatten."""
    x = mel.unsqueeze(1)
    for l in range(n_layers):
        p = "subsampling_module.layers.%d" % l
        x = F.conv2d(x, _t(sd, p + ".0.weight"), _t(sd, p + ".0.bias"), stride=2, padding=1)
        x = F.batch_norm(x, _t(sd, p + ".1.running_mean"), _t(sd, p + ".1.running_var"),
                         _t(sd, p + ".1.weight"), _t(sd, p + ".1.bias"), False, 0.0, BN_EPS)
        x = x * torch.sigmoid(x)                                   # Swish, activations.py:28-29
        if lens is not None:
            lens = torch.div(lens - 1, 2, rounding_mode="floor") + 1   # modules.py:243
    b, c, f, t = x.shape
    return x.reshape(b, c * f, t), lens                            # feature index = c*F' + f (modules.py:247)


# --------------------------------------------------------------------------
# a5  feed-forward module  (modules.py:385-395; half-step residual at blocks.py:122,132)
# --------------------------------------------------------------------------

def ffn(x: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), _t(sd, prefix + ".layers.0.weight"), _t(sd, prefix + ".layers.0.bias"), LN_EPS)
    h = F.linear(h, _t(sd, prefix + ".layers.1.weight"), _t(sd, prefix + ".layers.1.bias"))
    h = h * torch.sigmoid(h)
    return F.linear(h, _t(sd, prefix + ".layers.4.weight"), _t(sd, prefix + ".layers.4.bias"))


# --------------------------------------------------------------------------
# a6  (grouped) relative-position multi-head self-attention
#     attentions.py:549-620 (G=1), 645-718 (G>1), tables 1209-1257 / 1268-1315,
#     key-padding mask attentions.py:1326-1340, 1377-1403; wrapper modules.py:472-488
# --------------------------------------------------------------------------

def rel_sinusoid_rows(tp: int, dim: int, group: int) -> torch.Tensor:
    """R[m] = sinusoid(p = Tp-1-floor(G/2)-m), m in [0, 2Tp-G)  (the slice the reference takes of its
    precomputed table, attentions.py:1251 / 1309; same fp32 operation order as the table build 1219-1226)."""
    hi = tp - 1 - group // 2
    pos = torch.arange(hi, -hi - 1, -1, dtype=torch.float).unsqueeze(1)
    angles = pos / 10000 ** (2 * torch.arange(0, dim // 2, dtype=torch.float).unsqueeze(0) / dim)
    r = torch.zeros(pos.shape[0], dim)
    r[:, 0::2] = angles.sin()
    r[:, 1::2] = angles.cos()
    return r


def rel_sinusoid_rows_causal(tp: int, dim: int) -> torch.Tensor:
    """causal tables: R[m] = sinusoid(p = Tp-1-m), m in [0, Tp)  (attentions.py:1243-1247 / 1296-1300: the slice [max_len - T, max_len))"""
    pos = torch.arange(tp - 1, -1, -1, dtype=torch.float).unsqueeze(1)
    angles = pos / 10000 ** (2 * torch.arange(0, dim // 2, dtype=torch.float).unsqueeze(0) / dim)
    r = torch.zeros(pos.shape[0], dim)
    r[:, 0::2] = angles.sin()
    r[:, 1::2] = angles.cos()
    return r


def relpos_attention(x: torch.Tensor, lens: Optional[torch.Tensor], sd, prefix: str, heads: int, group: int,
                     return_probs: bool = False, causal: bool = False, left: Optional[int] = None, right: Optional[int] = None,
                     mask_stride: int = 1):
    """x: (B, T, D) *already pre-normed*; lens: valid frames per utterance at this stage.
    Closed form: S[b,h,i,j] = (Qu_i.K_j + Qv_i.E[Tg-1+j-i]) / sqrt(d).
    Streaming contexts (attentions.py:1377-1403, sliced ::s at encoders.py:132-136 and ::G at attentions.py:698): grouped key j is masked
    for grouped query i iff mask_stride * G * (j - i) > right or < -left.  causal (attentions.py:506-529, 1243-1247): E has Tg rows and the
    relative-to-absolute shift is the Music-Transformer skew; entries right of the diagonal are whatever the skew wraps in - the causal
    mask (right = 0) covers them."""
    bsz, t, dim = x.shape
    m = prefix + ".mhsa."
    q = F.linear(x, _t(sd, m + "query_layer.weight"), _t(sd, m + "query_layer.bias"))
    k = F.linear(x, _t(sd, m + "key_layer.weight"), _t(sd, m + "key_layer.bias"))
    v = F.linear(x, _t(sd, m + "value_layer.weight"), _t(sd, m + "value_layer.bias"))
    tp = (t + group - 1) // group * group                 # chunk padding, attentions.py:107-138, 671
    pad = tp - t
    if pad:
        q, k, v = (F.pad(z, (0, 0, 0, pad)) for z in (q, k, v))    # zeros *after* the projections
    qu = q + _t(sd, m + "u")                               # attentions.py:674-675 (pad rows become u / v)
    qv = q + _t(sd, m + "v")
    tg = tp // group
    d = group * dim // heads                               # attentions.py:643
    rows = rel_sinusoid_rows_causal(tp, dim) if causal else rel_sinusoid_rows(tp, dim, group)
    e = F.linear(rows, _t(sd, m + "pos_layer.weight"), _t(sd, m + "pos_layer.bias"))

    def split(z, rows):                                    # (B, rows*G, D) -> (B, H, rows, d): a pure view + transpose
        return z.reshape(z.shape[0], rows, heads, d).transpose(1, 2)
    qu, qv, k, v = split(qu, tg), split(qv, tg), split(k, tg), split(v, tg)
    e = split(e.unsqueeze(0), tg if causal else 2 * tg - 1)[0]               # (H, 2Tg-1, d); causal: (H, Tg, d)
    s_k = qu @ k.transpose(2, 3)                           # (B, H, Tg, Tg)
    s_rel = qv @ e.transpose(1, 2)                         # (B, H, Tg, 2Tg-1); causal: (B, H, Tg, Tg)
    i = torch.arange(tg).unsqueeze(1)
    j = torch.arange(tg).unsqueeze(0)
    if causal:      # attentions.py:506-529: pad one column left, flatten, pad the start, reshape (1 + Tg, Tg), drop the first row
        z = F.pad(s_rel, (1, 0)).reshape(bsz, heads, -1)
        z = F.pad(z, (0, 0)).reshape(bsz, heads, 1 + tg, tg)[:, :, 1:]       # seq_length2 - seq_length1 = 0 start padding
        s = (s_k + z) / d ** 0.5
    else:
        idx = (tg - 1 + j - i).expand(bsz, heads, tg, tg)      # rel_to_abs == this gather (attentions.py:483-547)
        s = (s_k + torch.gather(s_rel, 3, idx)) / d ** 0.5     # scale by grouped d, attentions.py:692
    if left is not None or right is not None:                  # streaming mask, on the positions the slicing keeps
        pos = torch.arange(tg) * (mask_stride * group)
        delta = pos.unsqueeze(0) - pos.unsqueeze(1)            # [i][j] = pos_j - pos_i
        band = torch.zeros(tg, tg)
        if right is not None:
            band = torch.maximum(band, (delta > right).float())
        if left is not None:
            band = torch.maximum(band, (delta < -left).float())
    else:
        band = None
    if lens is not None:
        # key group j masked iff its first frame G*j >= lens[b] (mask[:, :, ::G, ::G], attentions.py:698;
        # chunk padding is masked too, attentions.py:128-131); additive -1e9 as in the reference (:701)
        masked = (torch.arange(tg).unsqueeze(0) * group >= lens.unsqueeze(1)).float()[:, None, None, :]
        if band is not None:
            masked = torch.maximum(masked, band[None, None])   # streaming_mask.maximum(padding_mask), attentions.py:1399
        s = s + masked * -1e9
    elif pad or band is not None:
        masked = (torch.arange(tg) * group >= t).float()[None, None, None, :]
        if band is not None:
            masked = torch.maximum(masked, band[None, None])
        s = s + masked * -1e9
    p = s.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(bsz, tp, dim)[:, :t]      # un-group, drop chunk padding (:710-713)
    o = F.linear(o, _t(sd, m + "output_layer.weight"), _t(sd, m + "output_layer.bias"))
    return (o, p) if return_probs else o


def mhsa_module(x, lens, sd, prefix, heads, group, return_probs: bool = False, **ctx):
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), _t(sd, prefix + ".norm.weight"), _t(sd, prefix + ".norm.bias"), LN_EPS)  # modules.py:475
    return relpos_attention(h, lens, sd, prefix, heads, group, return_probs=return_probs, **ctx)


# --------------------------------------------------------------------------
# a7  convolution module  (modules.py:511-525; layers.py:122-136; activations.py:28-29, 37-39)
# --------------------------------------------------------------------------

def conv_module(x: torch.Tensor, sd, prefix: str, kernel: int, stride: int, causal: bool = False) -> torch.Tensor:
    d = x.shape[-1]
    p = prefix + ".layers"
    h = F.layer_norm(x, (d,), _t(sd, p + ".0.weight"), _t(sd, p + ".0.bias"), LN_EPS)
    h = F.linear(h, _t(sd, p + ".2.weight")[:, :, 0], _t(sd, p + ".2.bias"))         # pointwise conv == GEMM
    a, g = h.chunk(2, dim=-1)                                                         # GLU over channels
    h = (a * torch.sigmoid(g)).transpose(1, 2)                                        # (B, De, T)
    half = (kernel - 1) // 2
    h = F.pad(h, (kernel - 1, 0) if causal else (half, half))                         # "causal" / "same" pre-padding, layers.py:97-101
    h = F.conv1d(h, _t(sd, p + ".4.weight"), _t(sd, p + ".4.bias"), stride=stride, groups=h.shape[1])
    h = F.batch_norm(h, _t(sd, p + ".5.running_mean"), _t(sd, p + ".5.running_var"),
                     _t(sd, p + ".5.weight"), _t(sd, p + ".5.bias"), False, 0.0, BN_EPS)
    h = (h * torch.sigmoid(h)).transpose(1, 2)                                        # (B, To, De)
    return F.linear(h, _t(sd, p + ".7.weight")[:, :, 0], _t(sd, p + ".7.bias"))


# --------------------------------------------------------------------------
# a8  Conformer block  (blocks.py:119-137; residual paths blocks.py:99-114)
# --------------------------------------------------------------------------

def conformer_block(x, lens, sd, bp, trace: Optional[dict] = None, plan=None):
    """bp: efficientconformer_amd.config.BlockPlan.  Returns x_out (B, To, De).
    ``trace`` (optional) receives the four module outputs the reference's forward hooks see."""
    p = "blocks.%d" % bp.index
    f1 = ffn(x, sd, p + ".feed_forward_module1")
    x = x + 0.5 * f1
    x_ffn1 = x
    ctx = {}
    if plan is not None and (plan.causal or plan.left_context < (1 << 30) or plan.right_context < (1 << 30)):
        ctx = dict(causal=plan.causal, left=plan.left_context, right=plan.right_context, mask_stride=bp.mask_stride)
    att, att_w = mhsa_module(x, lens, sd, p + ".multi_head_self_attention_module", bp.num_heads, bp.group_size, return_probs=True, **ctx)
    x = x + att                                                   # att_res is Identity (att_stride == 1)
    x_mhsa = x
    c = conv_module(x, sd, p + ".convolution_module", bp.kernel_size, bp.conv_stride, causal=bool(plan is not None and plan.causal))
    if bp.transition:       # 1x1 strided conv on frames 0, s, 2s, ...   blocks.py:106-110
        res = F.linear(x[:, ::bp.conv_stride], _t(sd, p + ".conv_res.1.weight")[:, :, 0], _t(sd, p + ".conv_res.1.bias"))
    elif bp.conv_stride > 1:  # MaxPool1d(kernel 1, stride s) == frame decimation   blocks.py:110-114
        res = x[:, ::bp.conv_stride]
    else:
        res = x
    x = res + c
    x_conv = x
    f2 = ffn(x, sd, p + ".feed_forward_module2")
    x = x + 0.5 * f2
    x = F.layer_norm(x, (x.shape[-1],), _t(sd, p + ".norm.weight"), _t(sd, p + ".norm.bias"), LN_EPS)
    if trace is not None:
        trace[p + ".ffn1"], trace[p + ".mhsa"], trace[p + ".conv"], trace[p + ".ffn2"], trace[p + ".out"] = f1, att, c, f2, x
        trace[p + ".x_ffn1"], trace[p + ".x_mhsa"], trace[p + ".x_conv"] = x_ffn1, x_mhsa, x_conv   # residual stream
        trace[p + ".att_w"] = att_w        # the attention map the reference's forward returns per block (blocks.py:126, encoders.py:129)
    return x


# --------------------------------------------------------------------------
# a3/a4/a9  encoder shell  (encoders.py:97-142)
# --------------------------------------------------------------------------

def encoder_from_mel(mel: torch.Tensor, mel_len: Optional[torch.Tensor], sd, plan, trace: Optional[dict] = None):
    """mel (B, n_mels, Tm), lengths in mel frames -> (x (B, T_out, D_last), out_len)."""
    x, lens = subsample(mel, mel_len, sd, plan.sub_layers)
    if trace is not None:
        trace["subsample"] = x
    x = F.linear(x.transpose(1, 2), _t(sd, "linear.weight"), _t(sd, "linear.bias"))    # encoders.py:113-116
    if trace is not None:
        trace["linear"] = x
    for bp in plan.blocks:
        x = conformer_block(x, lens, sd, bp, trace, plan)
        if bp.conv_stride > 1 and lens is not None:
            lens = torch.div(lens - 1, bp.conv_stride, rounding_mode="floor") + 1     # encoders.py:139
    return x, lens


def encoder(audio: torch.Tensor, x_len: Optional[torch.Tensor], sd, plan):
    mel, mel_len = mel_frontend(audio, x_len, plan.n_fft, plan.win_length, plan.hop_length, plan.n_mels,
                                plan.sample_rate, plan.normalize, plan.mean, plan.std)
    return encoder_from_mel(mel, mel_len, sd, plan)


# --------------------------------------------------------------------------
# CTC head + greedy decode  (model_ctc.py:49, 57-68, 90-136)
# --------------------------------------------------------------------------

def ctc_logits(x: torch.Tensor, sd) -> torch.Tensor:
    return F.linear(x, _t(sd, "fc.weight"), _t(sd, "fc.bias"))


def ctc_greedy(logits: torch.Tensor, lens: torch.Tensor) -> List[List[int]]:
    """argmax per frame, drop blanks (id 0), collapse repeats not separated by a blank, stop at lens[b]
    (model_ctc.py:99-133; log_softmax does not change the argmax)."""
    preds = logits.argmax(dim=-1)
    out = []
    for b in range(preds.shape[0]):
        seq, prev = [], 0
        for t in range(int(lens[b])):
            c = int(preds[b, t])
            if c != 0 and c != prev:
                seq.append(c)
            prev = c
        out.append(seq)
    return out
