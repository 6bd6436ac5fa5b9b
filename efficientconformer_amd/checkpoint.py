"""Checkpoint import: load the reference's ``.ckpt`` files into the native models.

The reference saves ``{"model_state_dict", "optimizer_state_dict", "model_step", "tokenizer", "is_distributed"}`` with
``torch.save`` (reference models/model.py:345-359) and restores it in ``Model.load`` (model.py:361-384): DDP / DataParallel
checkpoints carry a ``.module.`` infix that is stripped when the loading model is not wrapped, the (pickled SentencePiece)
tokenizer object rides along.  Optimizer state and the scheduler step are training state and are ignored here.  The packed
native layout (bf16, padded, BatchNorm folded, K-permuted chain weights) is rebuilt from the fp32 tensors on first use.
"""
from __future__ import annotations

from typing import Union

import torch


def load_checkpoint(model, checkpoint: Union[str, dict], strict: bool = True):
    """``model.load(path)`` of the reference (model.py:361-384) for ``ModelCTC`` / ``Transducer``.

    ``checkpoint`` is a path or an already loaded dict.  Reference checkpoints pickle a tokenizer object, so the file is read
    with ``weights_only=False`` — only load checkpoints you trust.  A bare ``state_dict`` is accepted as well."""
    if isinstance(checkpoint, str):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    sd = checkpoint.get("model_state_dict", checkpoint) if isinstance(checkpoint, dict) else checkpoint
    if isinstance(checkpoint, dict) and checkpoint.get("is_distributed", False):
        sd = {k.replace(".module.", "."): v for k, v in sd.items()}                   # model.py:367-368
    result = model.load_state_dict(sd, strict=strict)
    if isinstance(checkpoint, dict) and checkpoint.get("tokenizer", None) is not None:
        model.tokenizer = checkpoint["tokenizer"]                                      # model.py:380
    return result


def save_checkpoint(model, path: str):
    """Inference-side counterpart of ``Model.save`` (model.py:345-359): same dict layout, no optimizer state."""
    torch.save({"model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "optimizer_state_dict": None, "model_step": 0, "tokenizer": getattr(model, "tokenizer", None),
                "is_distributed": False}, path)
