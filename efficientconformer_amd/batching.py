"""Batching front door: from single utterances on the host to padded batches resident in HBM.

* ``collate_fn_pad`` — the reference's collate (utils/preprocessing.py:27-45): sort by length descending, zero-pad.
* ``bucket_batches`` — length-bucketed batch plan: utterances sorted by length, cut into batches bounded by a number of
  utterances and by padded samples, so that padding waste (and the pad-frame work of the encoder) stays small.
* ``FrontDoor`` — runs a model over a list of waveforms: every batch is packed by worker threads into one of two persistent pinned
  host buffers and copied to the device on a side stream while the previous batch is being encoded; result ids come back through
  asynchronous downloads; results are returned in the caller's order.

The encoder's results for an utterance depend on the padded length of its batch (pad frames are live in the reference,
SURVEY.md section 8a); the plan is therefore deterministic for a given list of lengths.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import ctypes as C
import inspect
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib


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
    """Encode / decode a list of host waveforms with packing, H2D copies and result downloads overlapped with compute.

    Two kinds of per-batch function:

    * ``fn(audio (B, L) device tensor, lengths (B,) device tensor) -> list of per-utterance results`` - typically
      ``model.greedy_labels`` (``ModelCTC``) or ``model.greedy_tokens`` (``Transducer``).  It returns host lists, i.e. it synchronises
      once per batch; packing and the H2D copy of the next batch still overlap its GPU work.
    * ``device_fn(audio, lengths) -> (ids (B, N) int32 device tensor, counts (B,) int32 device tensor)`` - e.g.
      ``lambda x, n, host_n: model.encode_greedy(x, n, x_len_host=host_n)[2:]`` (the optional third argument is the batch's lengths as
      a host array) or ``lambda x, n: model.decode_encoded(*model.encoder(x, n)[:2])``.  Nothing
      synchronises inside the loop: the id tensors are downloaded asynchronously into pinned memory and turned into lists after the last
      batch, so the host packs batch k + 1 while the GPU still encodes batch k.

    The staging buffers are two pinned host buffers, allocated once and grown on demand (``tensor.pin_memory()`` per batch costs more
    than the batch's compute); rows are packed by ``workers`` native threads (``effconf_host_pack_rows``; a 256-utterance batch is 190 MB against
    5.6 ms of GPU work).  ``zero_pad = False`` skips zeroing the pad samples - for ragged batches, whose kernels never read them
    (``ConformerEncoder.ragged``); with the rectangular path the pad samples are live and must be zero."""

    def __init__(self, fn: Optional[Callable] = None, device="cuda", max_batch: int = 128, max_padded_samples: Optional[int] = None,
                 device_fn: Optional[Callable] = None, workers: int = 8, zero_pad: bool = True, pass_host_lengths: Optional[bool] = None):
        if (fn is None) == (device_fn is None):
            raise ValueError("FrontDoor needs exactly one of fn / device_fn")
        self.fn, self.device_fn, self.device = fn, device_fn, torch.device(device)
        self.max_batch, self.max_padded_samples = max_batch, max_padded_samples
        self.zero_pad = zero_pad
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._host = [None, None]                 # pinned fp32 staging, one per slot
        self._hlen = [None, None]
        self._free = [None, None]                 # event: the slot's last H2D copy has finished
        self._workers = max(1, workers)
        self._results: dict = {}
        self._stager = ThreadPoolExecutor(max_workers=1)
        # `pass_host_lengths` (explicit): device_fn(audio, lengths, host_lengths).  None = inferred: exactly three REQUIRED positional
        # parameters (a defaulted third argument or *args no longer receives the numpy lengths by accident)
        self._host_lengths = bool(pass_host_lengths)
        if device_fn is not None and pass_host_lengths is None:
            try:
                ps = list(inspect.signature(device_fn).parameters.values())
                req = [q for q in ps if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) and q.default is q.empty]
                self._host_lengths = len(req) == 3 and not any(q.kind == q.VAR_POSITIONAL for q in ps)
            except (TypeError, ValueError):
                pass
        self.stats: dict = {}                     # host seconds per piece of the last run() (pack, h2d_issue, launch, results)

    def _stage(self, waves: Sequence[torch.Tensor], idx: List[int], slot: int):
        t0 = time.perf_counter()
        # runs on the helper thread, whose current device is its own: pin_memory() below would otherwise create a context on GPU 0 in
        # every rank of a multi-GPU job (device = "cuda" without an index resolves to the copy stream's device)
        torch.cuda.set_device(self.copy_stream.device)
        lens = [int(waves[i].numel()) for i in idx]
        b, width = len(idx), max(max(lens), 1)
        if self._free[slot] is not None:
            self._free[slot].synchronize()        # the copy that last read this slot (two batches ago) is long done
        if self._host[slot] is None or self._host[slot].numel() < b * width:
            self._host[slot] = torch.empty(max(b * width, 1 << 20), dtype=torch.float32).pin_memory()
        if self._hlen[slot] is None or self._hlen[slot].numel() < b:
            self._hlen[slot] = torch.empty(max(b, 256), dtype=torch.int64).pin_memory()
        host = self._host[slot][:b * width].view(b, width)
        hlen = self._hlen[slot][:b]
        hlen.copy_(torch.tensor(lens, dtype=torch.int64))
        # native multi-threaded pack (csrc/hostpack.hip): per-row interpreter overhead and the GIL held 8 numpy threads at ~7 GB/s
        rows = []
        for r, i in enumerate(idx):
            w = waves[i].reshape(-1)
            if w.dtype != torch.float32 or not w.is_contiguous() or w.requires_grad or w.is_cuda:
                w = w.detach().to("cpu", torch.float32).contiguous()
            rows.append(w)                        # keeps converted rows alive across the call
        lib = _lib.load()
        ptrs = (C.c_void_p * b)(*[w.data_ptr() for w in rows])
        _lib.check(lib.effconf_host_pack_rows(ptrs, C.cast(hlen.data_ptr(), C.POINTER(C.c_int64)), b, host.data_ptr(), width,
                                              1 if self.zero_pad else 0, self._workers), "host_pack_rows")
        t1 = time.perf_counter()
        with torch.cuda.stream(self.copy_stream):
            dev = torch.empty(b, width, dtype=torch.float32, device=self.device)
            dlen = torch.empty(b, dtype=torch.int64, device=self.device)
            dev.copy_(host, non_blocking=True)
            dlen.copy_(hlen, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._free[slot] = ready
        t2 = time.perf_counter()
        self.stats["pack"] = self.stats.get("pack", 0.0) + t1 - t0
        self.stats["h2d_issue"] = self.stats.get("h2d_issue", 0.0) + t2 - t1
        return dev, dlen, ready, np.asarray(lens, dtype=np.int64)

    def close(self):
        """Stop the staging thread (idempotent); the pinned buffers are released with the object."""
        st, self._stager = self._stager, None
        if st is not None:
            st.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _result_buffer(self, key: int, n: int, dtype) -> torch.Tensor:
        # pinned download buffers, kept across runs (pin_memory() per batch costs 0.5 - 40 ms)
        buf = self._results.get(key)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=dtype).pin_memory()
            self._results[key] = buf
        return buf[:n]

    def run(self, waves: Sequence[torch.Tensor]) -> list:
        if self._stager is None:
            raise RuntimeError("FrontDoor.run after close()")
        plan = bucket_batches([int(w.numel()) for w in waves], self.max_batch, self.max_padded_samples)
        out: list = [None] * len(waves)
        self.stats = {}
        pending = []                              # device_fn: (idx, pinned ids, pinned counts) per batch
        # batch k + 1 is packed and copied by a helper thread WHILE this thread enqueues batch k's ~400 kernels (5 ms of host time per
        # 256-utterance batch against 6 ms on the GPU: staging on the same thread made the host the bottleneck)
        nxt_job = self._stager.submit(self._stage, waves, plan[0], 0) if plan else None
        cur = torch.cuda.current_stream(self.device)
        for k, idx in enumerate(plan):
            dev, dlen, ready, hl = nxt_job.result()
            nxt_job = self._stager.submit(self._stage, waves, plan[k + 1], (k + 1) & 1) if k + 1 < len(plan) else None
            cur.wait_event(ready)
            if self.device_fn is not None:
                # asynchronous; a three-argument device_fn also gets the lengths as a host array (ragged batches size their grids from
                # them: without it the encoder reads them back from the device - one synchronisation per batch)
                ids, counts = self.device_fn(dev, dlen, hl) if self._host_lengths else self.device_fn(dev, dlen)
                h_ids = self._result_buffer(2 * k, ids.numel(), ids.dtype).view(ids.shape)
                h_n = self._result_buffer(2 * k + 1, counts.numel(), counts.dtype).view(counts.shape)
                h_ids.copy_(ids, non_blocking=True)
                h_n.copy_(counts, non_blocking=True)
                pending.append((idx, h_ids, h_n))
            # both were allocated on the copy stream and are consumed on the caller's stream
            dev.record_stream(cur)
            dlen.record_stream(cur)
            if self.device_fn is None:
                res = self.fn(dev, dlen)
                for r, i in enumerate(idx):
                    out[i] = res[r]
        if pending:
            cur.synchronize()
            for idx, h_ids, h_n in pending:
                n = h_n.tolist()
                rows = h_ids.tolist()
                for r, i in enumerate(idx):
                    out[i] = rows[r][:n[r]]
        return out
