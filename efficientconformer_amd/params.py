"""State-dict surface of the encoder: key names, shapes and tensor roles.

The key names are part of the drop-in contract: a reference checkpoint's
``model_state_dict`` (keys ``encoder.<...>`` / ``fc.<...>``; reference
models/model.py:346-384, model_ctc.py:77-88) must load unchanged.  The names
follow the reference module tree (encoders.py:46-95, blocks.py:52-110,
modules.py:218-230, 376-383, 444-466, 498-509; attentions.py:57-60, 471-475).

``param_specs`` is the single source of truth used by the nn.Module container
(encoders.py in this package), the synthetic weight generator (synth.py) and
the C-ABI loader (which receives tensors by these key names).
"""
from __future__ import annotations

from typing import List, Tuple

from .config import EncoderPlan

# kinds: how synth.py initialises, and how the C packer interprets the tensor
W, B, GAMMA, BETA, RMEAN, RVAR, NBT, UV, EMB = "weight", "bias", "gamma", "beta", "running_mean", "running_var", "num_batches_tracked", "uv", "embedding"

Spec = Tuple[str, Tuple[int, ...], str]


def _ln(prefix, d) -> List[Spec]:
    return [(prefix + ".weight", (d,), GAMMA), (prefix + ".bias", (d,), BETA)]


def _bn(prefix, c) -> List[Spec]:
    return [(prefix + ".weight", (c,), GAMMA), (prefix + ".bias", (c,), BETA),
            (prefix + ".running_mean", (c,), RMEAN), (prefix + ".running_var", (c,), RVAR),
            (prefix + ".num_batches_tracked", (), NBT)]


def _lin(prefix, n_out, n_in) -> List[Spec]:
    return [(prefix + ".weight", (n_out, n_in), W), (prefix + ".bias", (n_out,), B)]


def _ffn(prefix, d, dff) -> List[Spec]:
    # Sequential(LN, Linear, Swish, Dropout, Linear, Dropout)  modules.py:376-383
    return _ln(prefix + ".layers.0", d) + _lin(prefix + ".layers.1", dff, d) + _lin(prefix + ".layers.4", d, dff)


def param_specs(plan: EncoderPlan) -> List[Spec]:
    """Ordered (key, shape, kind) list of the encoder's state_dict (no 'encoder.' prefix)."""
    out: List[Spec] = []
    cin = 1
    for l, c in enumerate(plan.sub_filters[:plan.sub_layers]):
        p = "subsampling_module.layers.%d" % l
        out += [(p + ".0.weight", (c, cin, 3, 3), W), (p + ".0.bias", (c,), B)] + _bn(p + ".1", c)
        cin = c
    out += _lin("linear", plan.blocks[0].dim_model, plan.dim_in)
    for b in plan.blocks:
        p = "blocks.%d" % b.index
        d, de = b.dim_model, b.dim_expand
        out += _ffn(p + ".feed_forward_module1", d, b.dim_ffn1)
        m = p + ".multi_head_self_attention_module"
        out += _ln(m + ".norm", d)
        out += [(m + ".mhsa.u", (d,), UV), (m + ".mhsa.v", (d,), UV)]
        for name in ("query_layer", "key_layer", "value_layer", "output_layer", "pos_layer"):
            out += _lin(m + ".mhsa." + name, d, d)
        c = p + ".convolution_module.layers"
        out += _ln(c + ".0", d)
        out += [(c + ".2.weight", (2 * de, d, 1), W), (c + ".2.bias", (2 * de,), B)]
        out += [(c + ".4.weight", (de, 1, b.kernel_size), W), (c + ".4.bias", (de,), B)]
        out += _bn(c + ".5", de)
        out += [(c + ".7.weight", (de, de, 1), W), (c + ".7.bias", (de,), B)]
        out += _ffn(p + ".feed_forward_module2", de, b.dim_ffn2)
        out += _ln(p + ".norm", de)
        if b.transition:
            out += [(p + ".conv_res.1.weight", (de, d, 1), W), (p + ".conv_res.1.bias", (de,), B)]
    return out


def head_specs(plan: EncoderPlan, vocab: int) -> List[Spec]:
    """CTC head ``fc`` (reference model_ctc.py:49)."""
    return _lin("fc", vocab, plan.dim_out)


def transducer_specs(dim_encoder: int, decoder_params: dict, joint_params: dict) -> List[Spec]:
    """Prediction + joint network state_dict (reference decoders.py:46-47 -> ``decoder.embedding`` / ``decoder.rnn``;
    joint_networks.py:41-52 -> ``joint_network.linear_{encoder,decoder,joint}``), RNN decoder, one layer."""
    v, h, j = int(decoder_params["vocab_size"]), int(decoder_params["dim_model"]), int(joint_params["dim_model"])
    out: List[Spec] = [("decoder.embedding.weight", (v, h), EMB)]
    out += [("decoder.rnn.weight_ih_l0", (4 * h, h), W), ("decoder.rnn.weight_hh_l0", (4 * h, h), W),
            ("decoder.rnn.bias_ih_l0", (4 * h,), B), ("decoder.rnn.bias_hh_l0", (4 * h,), B)]
    out += _lin("joint_network.linear_encoder", j, dim_encoder)
    out += _lin("joint_network.linear_decoder", j, h)
    out += _lin("joint_network.linear_joint", v, j)
    return out
