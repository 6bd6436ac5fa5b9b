"""Transducer (RNN-T) wrapper: encoder + prediction network + joint network with greedy decoding on the GPU
(reference models/transducer.py:52-186, models/decoders.py:41-70, models/joint_networks.py:33-104).

Keeps the reference's attribute names (``encoder``, ``decoder.embedding``, ``decoder.rnn``, ``joint_network.linear_*``),
state-dict keys and method names (``gready_search_decoding``, the reference's spelling).  The per-utterance Python loop
of the reference — one decoder call, one joint call and one ``.argmax()`` host sync per decision — runs as one
persistent HIP kernel per batch (effconf_rnnt_greedy).  Training (``forward`` over the full (T, U) lattice, RNN-T loss),
beam search and the LM fusion are out of scope (HISTORY.md).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .config import load_config
from .encoders import ConformerEncoder

_JOINT_MODES = {"sum": 0}
_JOINT_ACTS = {"tanh": 0}


class RnnDecoder(nn.Module):
    """Parameter container with the reference's key names (decoders.py:46-47); never executed in Python."""

    def __init__(self, params: dict):
        super().__init__()
        self.embedding = nn.Embedding(params["vocab_size"], params["dim_model"], padding_idx=0)
        self.rnn = nn.LSTM(input_size=params["dim_model"], hidden_size=params["dim_model"], num_layers=params["num_layers"],
                           batch_first=True, bidirectional=False)


class JointNetwork(nn.Module):
    """Parameter container with the reference's key names (joint_networks.py:41-52)."""

    def __init__(self, dim_encoder: int, dim_decoder: int, vocab_size: int, params: dict):
        super().__init__()
        assert params["act"] in ["tanh", "relu", "swish", None]
        assert params["joint_mode"] in ["concat", "sum"]
        if params.get("dim_model") is None or params["joint_mode"] not in _JOINT_MODES or params["act"] not in _JOINT_ACTS:
            raise NotImplementedError("native joint network: joint_mode 'sum', act 'tanh', dim_model set "
                                      "(every shipped Transducer config); got %r" % (params,))
        self.linear_encoder = nn.Linear(dim_encoder, params["dim_model"])
        self.linear_decoder = nn.Linear(dim_decoder, params["dim_model"])
        self.linear_joint = nn.Linear(params["dim_model"], vocab_size)
        self.joint_mode, self.act_name = params["joint_mode"], params["act"]


class Transducer(nn.Module):

    def __init__(self, encoder_params: dict, decoder_params: dict, joint_params: dict, tokenizer_params: Optional[dict] = None,
                 training_params: Optional[dict] = None, decoding_params: Optional[dict] = None, name: str = "model",
                 tokenizer=None):
        super().__init__()
        if encoder_params.get("arch", "Conformer") != "Conformer":
            raise Exception("Unknown encoder architecture:", encoder_params.get("arch"))
        if decoder_params.get("arch", "RNN") != "RNN":
            raise NotImplementedError("native prediction network: arch 'RNN' (every shipped Transducer config)")
        self.encoder = ConformerEncoder(encoder_params)
        self.decoder = RnnDecoder(decoder_params)
        self.joint_network = JointNetwork(self.encoder.plan.dim_out, decoder_params["dim_model"], decoder_params["vocab_size"],
                                          joint_params)
        self.max_consec_dec_step = decoder_params.get("max_consec_dec_step", 5)      # transducer.py:83
        self._cfg = (self.encoder.plan.dim_out, decoder_params["dim_model"], joint_params["dim_model"],
                     decoder_params["vocab_size"], decoder_params["num_layers"])
        self.tokenizer = tokenizer
        self.name = name
        self._rnnt = None
        self._rnnt_packed = False
        self.eval()

    @classmethod
    def from_config(cls, cfg, tokenizer=None):
        cfg = load_config(cfg)
        return cls(cfg["encoder_params"], cfg["decoder_params"], cfg["joint_params"], cfg.get("tokenizer_params"),
                   cfg.get("training_params"), cfg.get("decoding_params"), cfg.get("model_name", "model"), tokenizer)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # strip the DDP / DataParallel infix first (model.py:367-370: "encoder.module.preprocessing.*" in a raw DDP state dict),
        # then drop torchaudio's frontend buffers (the native frontend builds its own window / filterbank tables)
        sd = {k.replace(".module.", "."): v for k, v in state_dict.items()}
        sd = {k: v for k, v in sd.items() if not k.startswith("encoder.preprocessing.")}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.encoder.repack()
        self._rnnt_packed = False
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.encoder.repack()
        self._rnnt_packed = False
        return r

    def load(self, path):
        """Reference ``Model.load`` (model.py:361-384): a ``.ckpt`` path or dict; restores weights and the pickled tokenizer."""
        from .checkpoint import load_checkpoint
        return load_checkpoint(self, path)

    def forward(self, batch):
        raise NotImplementedError("Transducer.forward builds the (B, T, U+1, V) training lattice (transducer.py:88-107): "
                                  "training is out of scope of the native inference path; use greedy_tokens / gready_search_decoding")

    # ------------------------------------------------------------------ native handle
    def _ensure_rnnt(self):
        if self._rnnt_packed:
            return
        lib = _lib.load()
        if self._rnnt is not None:
            lib.effconf_rnnt_destroy(self._rnnt)
            self._rnnt = None
        de, h, j, v, layers = self._cfg
        cfg = _lib.EcRnntConfig(de, h, j, v, layers, int(self.max_consec_dec_step), _JOINT_MODES[self.joint_network.joint_mode],
                                _JOINT_ACTS[self.joint_network.act_name])
        handle = lib.effconf_rnnt_create(C.byref(cfg))
        if not handle:
            raise _lib.EffconfError("effconf_rnnt_create: %s" % lib.effconf_last_error().decode())
        self._rnnt = handle
        for prefix, mod in (("decoder.", self.decoder), ("joint_network.", self.joint_network)):
            for key, t in mod.state_dict().items():
                arr = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
                shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
                _lib.check(lib.effconf_rnnt_load_tensor(handle, (prefix + key).encode(), arr.ctypes.data_as(C.c_void_p), shape,
                                                        arr.ndim), "rnnt_load_tensor(%s)" % key)
        _lib.check(lib.effconf_rnnt_finalize(handle), "rnnt_finalize")
        self._rnnt_packed = True

    def __del__(self):
        try:
            if self._rnnt is not None and _lib._lib is not None:
                _lib._lib.effconf_rnnt_destroy(self._rnnt)
        except Exception:
            pass

    def set_decode_option(self, name: str, value: int):
        """Forward an option to the native decoder (effconf_rnnt_set_option), e.g. ``cluster_decode`` = -1 auto / 0 / 1."""
        self._ensure_rnnt()
        _lib.check(_lib.load().effconf_rnnt_set_option(self._rnnt, name.encode(), int(value)), "rnnt_set_option(%s)" % name)

    # ------------------------------------------------------------------ decoding
    def decode_encoded(self, f: torch.Tensor, f_len: Optional[torch.Tensor]):
        """Greedy RNN-T decode of encoder outputs f (B, T, Denc) fp32 on the GPU -> (tokens (B, max_tok) i32, token_len (B) i32)."""
        if not f.is_cuda:
            raise RuntimeError("efficientconformer_amd runs on a HIP device only (no CPU fallback)")
        with torch.cuda.device(f.device):             # the C library allocates / launches on the current device
            return self._decode_encoded(f, f_len)

    def _decode_encoded(self, f: torch.Tensor, f_len: Optional[torch.Tensor]):
        self._ensure_rnnt()
        lib = _lib.load()
        f = f.contiguous().float()
        b, t, _ = f.shape
        if f_len is None:
            f_len = torch.full((b,), t, dtype=torch.int64, device=f.device)
        f_len = f_len.to(device=f.device, dtype=torch.int64).contiguous()
        max_tok = int(lib.effconf_rnnt_max_tokens(self._rnnt, t))
        tokens = torch.empty(b, max_tok, dtype=torch.int32, device=f.device)
        token_len = torch.empty(b, dtype=torch.int32, device=f.device)
        ws = torch.empty(int(lib.effconf_rnnt_workspace_bytes(self._rnnt, b, t)), dtype=torch.uint8, device=f.device)
        _lib.check(lib.effconf_rnnt_greedy(self._rnnt, f.data_ptr(), f_len.data_ptr(), b, t, tokens.data_ptr(), token_len.data_ptr(),
                                           max_tok, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(f.device).cuda_stream),
                   "rnnt_greedy")
        return tokens, token_len

    def greedy_tokens(self, x: torch.Tensor, x_len: Optional[torch.Tensor], from_mel: bool = False) -> List[List[int]]:
        """Greedy token-id sequences (without the start token), one list per utterance."""
        f, f_len, _ = self.encoder.forward_mel(x, x_len) if from_mel else self.encoder(x, x_len)
        tokens, token_len = self.decode_encoded(f, f_len)
        tokens, token_len = tokens.cpu(), token_len.cpu()          # one D2H copy per batch
        return [tokens[i, :int(token_len[i])].tolist() for i in range(tokens.shape[0])]

    def gready_search_decoding(self, x, x_len):
        """Reference spelling (transducer.py:139).  Decoded strings when a tokenizer is attached
        (``tokenizer.decode(y[:, 1:].tolist())``, transducer.py:179), otherwise the id lists."""
        ids = self.greedy_tokens(x, x_len)
        return self.tokenizer.decode(ids) if self.tokenizer is not None else ids

    greedy_search_decoding = gready_search_decoding
