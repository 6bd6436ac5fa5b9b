"""CTC model wrapper: the caller side of the hot path (reference models/model_ctc.py:39-136).

``ModelCTC`` keeps the reference's attribute names (``encoder``, ``fc``), state-dict keys
(``encoder.*``, ``fc.*``) and method names (including the reference's spelling
``gready_search_decoding``), but the head and the greedy collapse run as HIP kernels
(effconf_ctc_greedy) instead of ``nn.Linear`` + a Python loop with ``.item()`` per token.
Training (losses, optimizer, schedules), beam search and WER scoring are out of scope (HISTORY.md).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from .config import load_config
from .encoders import ConformerEncoder


class ModelCTC(nn.Module):

    def __init__(self, encoder_params: dict, tokenizer_params: dict, training_params: Optional[dict] = None,
                 decoding_params: Optional[dict] = None, name: str = "model", tokenizer=None):
        super().__init__()
        if encoder_params.get("arch", "Conformer") != "Conformer":
            raise Exception("Unknown encoder architecture:", encoder_params.get("arch"))
        self.encoder = ConformerEncoder(encoder_params)
        self.fc = nn.Linear(self.encoder.plan.dim_out, tokenizer_params["vocab_size"])
        self.encoder.attach_head(self.fc)
        self.tokenizer = tokenizer
        self.name = name
        self.eval()

    @classmethod
    def from_config(cls, cfg, tokenizer=None):
        cfg = load_config(cfg)
        return cls(cfg["encoder_params"], cfg["tokenizer_params"], cfg.get("training_params"),
                   cfg.get("decoding_params"), cfg.get("model_name", "model"), tokenizer)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # strip the DDP / DataParallel infix first (model.py:367-370: "encoder.module.preprocessing.*" in a raw DDP state dict),
        # then drop torchaudio's frontend buffers (the native frontend builds its own window / filterbank tables)
        sd = {k.replace(".module.", "."): v for k, v in state_dict.items()}
        sd = {k: v for k, v in sd.items() if not k.startswith("encoder.preprocessing.")}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.encoder.repack()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.encoder.repack()
        return r

    def load(self, path):
        """Reference ``Model.load`` (model.py:361-384): a ``.ckpt`` path or dict; restores weights and the pickled tokenizer."""
        from .checkpoint import load_checkpoint
        return load_checkpoint(self, path)

    # ---- reference ModelCTC.forward (model_ctc.py:57-68): batch = (x, y, x_len, y_len)
    def forward(self, batch, return_attentions: bool = False):
        x, _, x_len, _ = batch
        enc, enc_len, attentions = self.encoder(x, x_len, return_attentions=return_attentions)
        logits, _, _ = self._head(enc, enc_len, want_logits=True)
        return logits, enc_len, attentions

    def _head(self, enc: torch.Tensor, enc_len: Optional[torch.Tensor], want_logits: bool = False):
        lib = _lib.load()
        if not enc.is_cuda:
            raise RuntimeError("efficientconformer_amd runs on a HIP device only (no CPU fallback)")
        # bf16 rows (a gathered chunk of the multi-rank path on a bf16 wire) go to the head as they are: effconf_ctc_greedy_bf16 - no widening pass,
        # two MFMAs per 16 k, labels identical to the fp32-input head on the same values
        # (only while the handle's head IS the split-bf16 kernel, option ctc_mfma = 2, the default: with 0 / 1 the fp32 head is a k-ordered fp32 chain and the
        # bf16 rows are widened to it, so gathered and local chunks keep producing the same labels)
        as_bf16 = enc.dtype == torch.bfloat16 and self.encoder.precision == "bf16" and self.encoder._options.get("ctc_mfma", 2) == 2
        enc = enc.contiguous() if as_bf16 else enc.contiguous().float()
        b, t, _ = enc.shape
        if enc_len is None:
            enc_len = torch.full((b,), t, dtype=torch.int64, device=enc.device)
        enc_len = enc_len.to(enc.device, torch.int64).contiguous()
        labels = torch.empty(b, t, dtype=torch.int32, device=enc.device)
        label_len = torch.empty(b, dtype=torch.int32, device=enc.device)
        logits = torch.empty(b, t, self.fc.out_features, dtype=torch.float32, device=enc.device) if want_logits else None
        ws = torch.empty(b * t * 4, dtype=torch.uint8, device=enc.device)
        with torch.cuda.device(enc.device):          # the C library launches on the current device
            self.encoder._ensure_packed()
            fn = lib.effconf_ctc_greedy_bf16 if as_bf16 else lib.effconf_ctc_greedy
            _lib.check(fn(self.encoder._handle, enc.data_ptr(), enc_len.data_ptr(), b, t, labels.data_ptr(),
                          label_len.data_ptr(), logits.data_ptr() if want_logits else None, ws.data_ptr(),
                          ws.numel(), torch.cuda.current_stream(enc.device).cuda_stream), "ctc_greedy")
        return logits, labels, label_len

    def encode_greedy(self, x: torch.Tensor, x_len: Optional[torch.Tensor], from_mel: bool = False, **encoder_kwargs):
        """Encoder + CTC head with the head launched PER ROW RANGE on that range's stream (``ConformerEncoder.sub_batches``): the
        fc + argmax + collapse of range i overlaps the other ranges' encoder kernels instead of running after the join
        (reference: model_ctc.py:90-133 runs them one after the other).  Returns (enc, enc_len, labels, label_len); the labels are
        the same as ``_head(enc, enc_len)`` on the joined output - the head is row-local."""
        lib = _lib.load()
        state = {}

        def hook(lo, hi, out, out_len):
            if "labels" not in state:
                b, t = out.shape[0], out.shape[1]
                state["labels"] = torch.empty(b, t, dtype=torch.int32, device=out.device)
                state["label_len"] = torch.empty(b, dtype=torch.int32, device=out.device)
            t = out.shape[1]
            st = torch.cuda.current_stream(out.device)
            ws = torch.empty((hi - lo) * t * 4, dtype=torch.uint8, device=out.device)
            _lib.check(lib.effconf_ctc_greedy(self.encoder._handle, out[lo:].data_ptr(), out_len[lo:].data_ptr(), hi - lo, t,
                                              state["labels"][lo:].data_ptr(), state["label_len"][lo:].data_ptr(), None, ws.data_ptr(),
                                              ws.numel(), st.cuda_stream), "ctc_greedy")
            for tns in (state["labels"], state["label_len"]):
                tns.record_stream(st)

        with torch.cuda.device(x.device):
            enc, enc_len, _ = (self.encoder.forward_mel if from_mel else self.encoder)(x, x_len, range_hook=hook, **encoder_kwargs)
        return enc, enc_len, state["labels"], state["label_len"]

    def greedy_labels(self, x: torch.Tensor, x_len: Optional[torch.Tensor], from_mel: bool = False) -> List[List[int]]:
        """Greedy CTC label-id sequences (blank 0 removed, repeats collapsed), one list per utterance."""
        enc, enc_len, _ = self.encoder.forward_mel(x, x_len) if from_mel else self.encoder(x, x_len)
        _, labels, label_len = self._head(enc, enc_len)
        labels, label_len = labels.cpu(), label_len.cpu()          # one D2H copy per batch, not one per token
        return [labels[b, :int(label_len[b])].tolist() for b in range(labels.shape[0])]

    def gready_search_decoding(self, x, x_len):
        """Reference spelling (model_ctc.py:90).  Returns decoded strings when a tokenizer is attached
        (``tokenizer.decode(list_of_id_lists)``, model_ctc.py:136), otherwise the id lists."""
        ids = self.greedy_labels(x, x_len)
        return self.tokenizer.decode(ids) if self.tokenizer is not None else ids

    greedy_search_decoding = gready_search_decoding
