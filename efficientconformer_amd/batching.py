"""Batching front door: from single utterances on the host to padded batches resident in HBM.

* ``collate_fn_pad`` — the reference's collate (utils/preprocessing.py:27-45): sort by length descending, zero-pad.
* ``bucket_batches`` — length-bucketed batch plan: utterances sorted by length, cut into batches bounded by a number of
  utterances and by padded samples, so that padding waste (and the pad-frame work of the encoder) stays small.
* ``FrontDoor`` — runs a model over a list of waveforms: every batch is assembled in a pinned host buffer, copied to the device
  on a side stream while the previous batch is being encoded (double buffering), results returned in the caller's order.

The encoder's results for an utterance depend on the padded length of its batch (pad frames are live in the reference,
SURVEY.md section 8a); the plan is therefore deterministic for a given list of lengths.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch


def collate_fn_pad(batch):
    """Reference ``collate_fn_pad`` (utils/preprocessing.py:27-45), regular mode and an audio-only inference mode.

    batch: list of ``[audio (1, L) tensor, label (U,) tensor]`` or of ``[audio]`` / bare audio tensors.
    Returns ``(data, target, data_lengths, target_lengths)`` (targets ``None`` in audio-only mode), sorted by length descending."""
    items = [b if isinstance(b, (list, tuple)) else [b] for b in batch]
    order = sorted(range(len(items)), key=lambda i: items[i][0].shape[-1], reverse=True)
    data = [items[i][0].reshape(-1) for i in order]
    data_lengths = torch.tensor([len(d) for d in data], dtype=torch.long)
    data = torch.nn.utils.rnn.pad_sequence(data, batch_first=True, padding_value=0)
    if len(items[0]) < 2:
        return data, None, data_lengths, None
    target = [items[i][1] for i in order]
    target_lengths = torch.tensor([t.size(0) for t in target], dtype=torch.long)
    target = torch.nn.utils.rnn.pad_sequence(target, batch_first=True, padding_value=0)
    return data, target, data_lengths, target_lengths


def bucket_batches(lengths: Sequence[int], max_batch: int = 128, max_padded_samples: Optional[int] = None) -> List[List[int]]:
    """Indices of ``lengths`` grouped into batches: sorted by length descending (ties by index), greedily cut when a batch
    would exceed ``max_batch`` utterances or ``max_padded_samples`` = utterances x longest length."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    batches: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        longest = int(lengths[cur[0]]) if cur else int(lengths[i])
        if cur and (len(cur) >= max_batch or (max_padded_samples is not None and (len(cur) + 1) * longest > max_padded_samples)):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


class FrontDoor:
    """Encode / decode a list of host waveforms with H2D copies overlapped with compute.

    ``fn(audio (B, L) device tensor, lengths (B,) device tensor) -> list of per-utterance results`` is typically
    ``model.greedy_labels`` (``ModelCTC``) or ``model.greedy_tokens`` (``Transducer``)."""

    def __init__(self, fn: Callable, device, max_batch: int = 128, max_padded_samples: Optional[int] = None):
        self.fn, self.device = fn, torch.device(device)
        self.max_batch, self.max_padded_samples = max_batch, max_padded_samples
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _stage(self, waves: Sequence[torch.Tensor], idx: List[int]):
        lens = [int(waves[i].numel()) for i in idx]
        host = torch.zeros(len(idx), max(lens), dtype=torch.float32).pin_memory()
        for r, i in enumerate(idx):
            host[r, :lens[r]] = waves[i].reshape(-1)
        hlen = torch.tensor(lens, dtype=torch.int64).pin_memory()
        with torch.cuda.stream(self.copy_stream):
            dev = host.to(self.device, non_blocking=True)
            dlen = hlen.to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return dev, dlen, ready, (host, hlen)

    def run(self, waves: Sequence[torch.Tensor]) -> list:
        plan = bucket_batches([int(w.numel()) for w in waves], self.max_batch, self.max_padded_samples)
        out: list = [None] * len(waves)
        staged = self._stage(waves, plan[0]) if plan else None
        for k, idx in enumerate(plan):
            dev, dlen, ready, keep = staged
            nxt = self._stage(waves, plan[k + 1]) if k + 1 < len(plan) else None       # copy of batch k+1 overlaps compute of k
            torch.cuda.current_stream(self.device).wait_event(ready)
            res = self.fn(dev, dlen)
            # both were allocated on the copy stream and are consumed on the caller's stream
            dev.record_stream(torch.cuda.current_stream(self.device))
            dlen.record_stream(torch.cuda.current_stream(self.device))
            for r, i in enumerate(idx):
                out[i] = res[r]
            staged = nxt
        return out
