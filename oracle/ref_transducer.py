"""ORACLE — test infrastructure only, never part of the product path.

CPU (torch fp32) restatement of the RNN-T greedy decode of burchim/EfficientConformer: prediction network
(Embedding + 1-layer LSTM), joint network ("sum" + tanh) and the greedy loop with ``max_consec_dec_step``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Pinning: token sequences are pinned against the reference's own ``Transducer.gready_search_decoding`` executed in
the build container on key-seeded weights (tools/make_goldens.py -> tests/golden/rnnt_*.npz;
tests/test_oracle_golden.py).

``sd`` holds the reference's keys ``decoder.*`` / ``joint_network.*`` (efficientconformer_amd/params.py).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def _t(sd, key) -> torch.Tensor:
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def lstm_step(sd: Dict, y: int, h: torch.Tensor, c: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """One step of RnnDecoder.forward on a single token (reference models/decoders.py:52-67: Embedding lookup ->
    nn.LSTM, 1 layer; torch gate order i, f, g, o; c' = f*c + i*g; h' = o*tanh(c'))."""
    x = _t(sd, "decoder.embedding.weight")[y]
    g = (_t(sd, "decoder.rnn.weight_ih_l0") @ x + _t(sd, "decoder.rnn.bias_ih_l0")
         + _t(sd, "decoder.rnn.weight_hh_l0") @ h + _t(sd, "decoder.rnn.bias_hh_l0"))
    hd = h.numel()
    i, f, gg, o = torch.sigmoid(g[:hd]), torch.sigmoid(g[hd:2 * hd]), torch.tanh(g[2 * hd:3 * hd]), torch.sigmoid(g[3 * hd:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def joint_logits(sd: Dict, f_t: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """JointNetwork.forward in decoding mode (reference models/joint_networks.py:80-104, joint_mode 'sum', act tanh):
    linear_joint(tanh(linear_encoder(f) + linear_decoder(g)))."""
    fe = _t(sd, "joint_network.linear_encoder.weight") @ f_t + _t(sd, "joint_network.linear_encoder.bias")
    gd = _t(sd, "joint_network.linear_decoder.weight") @ g + _t(sd, "joint_network.linear_decoder.bias")
    return _t(sd, "joint_network.linear_joint.weight") @ torch.tanh(fe + gd) + _t(sd, "joint_network.linear_joint.bias")


def greedy_decode(sd: Dict, f: torch.Tensor, f_len, max_consec_dec_step: int = 5, with_margins: bool = False):
    """Transducer.gready_search_decoding (reference models/transducer.py:139-186) on encoder outputs f (B, T, Denc).

    Per utterance: start token 0 and zero LSTM state (151-152); the decoder advances only after an emitted token (159);
    a decision is the argmax of the joint logits (164: softmax().log().argmax() has the same argmax); blank (0) or
    ``consec == max_consec_dec_step`` moves to the next encoder frame and resets the counter (167-169), anything else is
    appended and the decoder runs again on the SAME frame (171-175).  Returns y[1:] per utterance (179), and optionally
    the smallest top-2 logit margin seen per utterance."""
    out: List[List[int]] = []
    margins: List[float] = []
    hd = _t(sd, "decoder.rnn.weight_hh_l0").shape[1]
    with torch.no_grad():
        for b in range(f.shape[0]):
            y: List[int] = [0]
            h, c = torch.zeros(hd), torch.zeros(hd)
            enc_step, consec, worst = 0, 0, float("inf")
            n = int(f_len[b])
            while enc_step < n:
                h, c = lstm_step(sd, y[-1], h, c)
                while enc_step < n:
                    logits = joint_logits(sd, f[b, enc_step], h)
                    pred = int(logits.argmax())
                    if with_margins:
                        top = logits.topk(2).values
                        worst = min(worst, float(top[0] - top[1]))
                    if pred == 0 or consec == max_consec_dec_step:
                        consec = 0
                        enc_step += 1
                    else:
                        consec += 1
                        y.append(pred)
                        break
            out.append(y[1:])
            margins.append(worst)
    return (out, margins) if with_margins else out
