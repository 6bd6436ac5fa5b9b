"""Multi-GPU: utterances shard data-parallel, one process per GPU, RCCL all-gather of encoder outputs.

The reference's only parallelism is data parallelism (DDP + DistributedSampler, reference main.py:33-35, 217-220,
model_ctc.py:70-75, functions.py:167-170); in eval mode its encoder performs no collective at all and only
decoded strings are gathered (model.py:463-466).  The path shards naturally along the batch: utterance b's
output depends only on its own row and on the padded length (pad frames are live), so

  * rank r takes rows r::world of the length-sorted global batch (round-robin keeps shards length-balanced);
  * every shard keeps the GLOBAL padded length, so results are bit-equal to the unsharded run;
  * every shard has ceil(B / world) rows (a short shard is filled with a copy of its last row), so all ranks
    launch identical shapes and every collective is one fixed-size all_gather_into_tensor (the RCCL fast path);
  * the encoder outputs are all-gathered PER SUB-BATCH ROW RANGE (`backend="nccl"` is RCCL over xGMI on ROCm):
    `ConformerEncoder.forward(..., range_hook=...)` calls back as soon as a row range's last kernel is enqueued, with
    that range's HIP stream current, and the collective is issued right there: the process group runs it on ITS OWN
    side stream behind the range's kernels, the other ranges' streams keep computing under it, and the range's stream
    (and whatever the consumer enqueues on it) continues once the chunk has arrived.  No further stream is created:
    on the MI355X more than four concurrently active HIP streams per process stop overlapping (a 5.5 ms step became
    7.4 ms with an extra comm + head stream; tools/overlap_probe.py), so the budget is: caller's stream + two side
    streams for the ranges + the process group's;
  * consumers (CTC head, RNN-T decode) work chunk by chunk on the gathered tensors (`Gathered.chunks`), or call
    `Gathered.assemble()` for one (B, T, D) tensor in global order;
  * `pipelined=True` (throughput serving): a range's collective is issued asynchronously and its consumer runs ONE CALL LATER, on the same
    range's stream right after that call's encoder kernels - so collective k is on the wire under the WHOLE encoder of call k + 1 instead
    of the tail of call k (ranges cut for equal work finish together: one GPU measured 24 % of the collectives' time covered, tools/
    overlap_probe.py).  Results arrive one call late; `flush()` drains the last call;
  * weights are replicated (<= 251 MB bf16), there is no other data-path collective;
  * HARDWARE QUEUES: run multi-rank processes with `GPU_MAX_HW_QUEUES=8` in the environment BEFORE the HIP runtime initialises (the first
    torch.cuda call).  HIP multiplexes streams onto that many hardware queues (default 4); the forward's three streams plus what an RCCL
    process group brings exceed four, and streams that share a queue serialise: measured on RCCL with one rank (`bench.py --force-dist`), a rank
    that merely holds a process group steps in 7.2 ms instead of 5.5 with four queues and in 5.55 ms with eight (profiles/r4_02_rccl_one_rank.txt).
    `bench.py` sets it for its rank processes; `ShardedEncoder` warns when it is missing.
"""
from __future__ import annotations

import inspect
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def _all_gather(dst: torch.Tensor, src: torch.Tensor, group=None):
    """Fixed-size all-gather into one tensor.  RCCL ("nccl") and gloo-on-CPU take `all_gather_into_tensor`; gloo with device tensors (a
    2-rank test on ONE GPU) goes through host copies."""
    if src.is_cuda and dist.get_backend(group) == "gloo":
        host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(host, src.cpu(), group=group)
        dst.copy_(torch.cat(host, 0).view(dst.shape), non_blocking=False)
    else:
        dist.all_gather_into_tensor(dst, src, group=group)


def shard_rows(batch: int, rank: int, world: int) -> torch.Tensor:
    return torch.arange(rank, batch, world) if rank < batch else torch.empty(0, dtype=torch.int64)


def shard_size(batch: int, world: int) -> int:
    return (batch + world - 1) // world


def shard_batch(x: torch.Tensor, x_len: torch.Tensor, rank: int, world: int, uniform: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rows rank::world of a (B, L) batch; L (the global pad length) is preserved on purpose.
    uniform=True: ceil(B / world) rows on every rank (a short shard repeats its last row; the copy is dropped after the gather)."""
    idx = shard_rows(x.shape[0], rank, world)
    if uniform:
        per = shard_size(x.shape[0], world)
        fill = idx[-1:] if len(idx) else torch.zeros(1, dtype=torch.int64)
        idx = torch.cat([idx] + [fill] * (per - len(idx)))
    idx = idx.to(x.device)
    return x.index_select(0, idx).contiguous(), x_len.index_select(0, idx).contiguous()


class GatheredChunk:
    """Local rows [lo, hi) of every rank: `out` is (world * (hi - lo), T, D) in wire dtype, row r * (hi - lo) + j  <->  global row
    (lo + j) * world + r (`rows`, entries >= global_batch are shard fill and are dropped by `keep`)."""

    def __init__(self, lo, hi, out, out_len, rows, keep, event, comm, works=None):
        self.lo, self.hi, self.out, self.out_len, self.rows, self.keep = lo, hi, out, out_len, rows, keep
        self._event, self._comm, self._works = event, comm, works

    def wait(self, stream: Optional["torch.cuda.Stream"] = None):
        """Make `stream` (default: the current stream) wait for this chunk's collective; keeps the buffers alive for it."""
        if self._works:                     # asynchronous collectives (pipelined mode): Work.wait() makes the CURRENT stream wait (RCCL)
            works, self._works = self._works, None
            if stream is not None and self.out.is_cuda:
                with torch.cuda.stream(stream):
                    for w in works:
                        w.wait()
            else:
                for w in works:
                    w.wait()
            if self.out.is_cuda:
                st = stream or torch.cuda.current_stream(self.out.device)
                self.out.record_stream(st)
                self.out_len.record_stream(st)
            return self
        if self._event is None:
            return self
        stream = stream or torch.cuda.current_stream(self.out.device)
        stream.wait_event(self._event)
        self.out.record_stream(stream)
        self.out_len.record_stream(stream)
        return self


class Gathered:
    def __init__(self, chunks: List[GatheredChunk], global_batch: int):
        self.chunks, self.global_batch = chunks, global_batch

    def wait(self, stream=None):
        for c in self.chunks:
            c.wait(stream)
        return self

    def assemble(self, dtype: Optional[torch.dtype] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """One (B, T, D) tensor + (B,) lengths in global row order (waits for every chunk on the current stream)."""
        self.wait()
        c0 = self.chunks[0]
        out = c0.out.new_empty((self.global_batch,) + tuple(c0.out.shape[1:]), dtype=dtype or c0.out.dtype)
        out_len = c0.out_len.new_empty(self.global_batch)
        for c in self.chunks:
            out[c.rows[c.keep]] = c.out[c.keep].to(out.dtype)
            out_len[c.rows[c.keep]] = c.out_len[c.keep]
        return out, out_len


class ShardedEncoder:
    """encoder(x, x_len) on this rank's shard + per-row-range all-gather on a comm stream (CUDA), global order restored.

    `encoder` is a `ConformerEncoder` (its `forward` accepts `range_hook`) or any callable `(x, x_len) -> (out, out_len, ...)`
    (one range; CPU / gloo tests wrap the oracle this way).  `wire_dtype=None` sends the encoder's fp32 outputs unchanged, so
    what a consumer computes from a gathered chunk is bit-identical to what the producing rank would compute locally;
    `torch.bfloat16` halves the xGMI bytes."""

    def close(self):
        """Pipelined: run the consumers of the last call.  (The wrapped encoder is never left modified: its ragged cut policy is switched to the
        rank-independent "rows" cut only for the duration of one `encode_shard` call - round 5 set it in the constructor and restored it from
        `__del__`, so a dropped wrapper could reset the policy under a live one: mismatched collectives.)"""
        if self._pending:
            self.flush()

    def __del__(self):
        try:
            if getattr(self, "_pending", None):
                import warnings
                warnings.warn("efficientconformer_amd.dist.ShardedEncoder dropped with %d gathered chunk(s) whose consumer never ran: call flush() "
                              "(or close()) after the last pipelined forward" % len(self._pending))
        except Exception:
            pass

    def __init__(self, encoder: Callable, group=None, wire_dtype: Optional[torch.dtype] = None, pipelined: bool = False):
        self.encoder, self.group, self.wire_dtype = encoder, group, wire_dtype
        self.pipelined = bool(pipelined)
        self._pending = []            # pipelined: (call id, (lo, hi), chunk, consumer) of collectives whose consumer has not run yet
        self._call = 0
        self._maps = {}
        # Every row range ends in ONE fixed-size collective, so all ranks must cut the SAME ranges.  A ragged ConformerEncoder cuts its
        # ranges for equal valid frames of the lengths it is handed - different on every rank (mismatched collectives: a hang or a
        # corrupted gather).  Under this class it cuts by row count instead (rank-independent: every shard has the same number of rows),
        # unless the caller pins explicit `sub_batch_bounds` (bench.py: frame-balanced / staggered cuts computed from lengths all ranks know).
        # The policy is asserted PER CALL (encode_shard sets "rows" around the encoder call and restores what it found), never left on the shared encoder:
        # any number of wrappers of one encoder may come and go in any order.
        import os
        import warnings
        try:
            nq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        except ValueError:
            nq = 4
        if nq < 8 and torch.cuda.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
            warnings.warn("efficientconformer_amd.dist: GPU_MAX_HW_QUEUES=%d - with an RCCL process group the forward's streams share hardware queues "
                          "and serialise (+30 %% per step measured); export GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises" % nq)
        fwd = getattr(encoder, "forward", encoder)
        try:
            self._hooked = "range_hook" in inspect.signature(fwd).parameters
        except (TypeError, ValueError):
            self._hooked = False

    # ------------------------------------------------------------------ collective of one row range
    def _gather_range(self, lo: int, hi: int, out: torch.Tensor, out_len: torch.Tensor, global_batch: int, chunks: list, consumer=None):
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n = hi - lo
        # row map of the gathered chunk, built ON the output's device and cached per (lo, hi, world, global batch): a pageable
        # host -> device copy here would make PyTorch synchronise the range's compute stream, i.e. block the host until range i's whole
        # encoder has finished before range i + 1 is enqueued (the overlap this class exists for)
        key = (lo, hi, world, global_batch, str(out.device))
        if key not in self._maps:
            rows = ((torch.arange(lo, hi, device=out.device).view(1, n)) * world + torch.arange(world, device=out.device).view(world, 1)).reshape(-1)
            self._maps[key] = (rows, rows < global_batch)
        rows, keep = self._maps[key]
        if out.is_cuda:
            dev = out.device
            # issued on the range's own stream (current): RCCL's process-group stream waits for this stream's kernels, the other ranges
            # compute under the transfer, this stream resumes when the chunk is there
            wire = out[lo:hi] if self.wire_dtype is None else out[lo:hi].to(self.wire_dtype)
            wl = out_len[lo:hi]
            g = wire.new_empty((world * n,) + tuple(wire.shape[1:]))
            gl = wl.new_empty(world * n)
            if self.pipelined:
                # the PREVIOUS call's chunk of this row range: its collective ran under this call's encoder kernels and is (long) done;
                # its consumer runs here, on the range's stream, behind this call's encoder and in front of this call's collective
                self._consume_pending((lo, hi))
                works = None
                if dist.get_backend(self.group) == "gloo":        # device tensors over gloo (one-GPU tests): host copies, synchronous
                    _all_gather(g, wire.contiguous(), self.group)
                    _all_gather(gl, wl.contiguous(), self.group)
                else:
                    works = [dist.all_gather_into_tensor(g, wire.contiguous(), group=self.group, async_op=True),
                             dist.all_gather_into_tensor(gl, wl.contiguous(), group=self.group, async_op=True)]
                ch = GatheredChunk(lo, hi, g, gl, rows, keep, None, torch.cuda.current_stream(dev), works)
                chunks.append(ch)
                self._pending.append((self._call, (lo, hi), ch, consumer))
                return
            _all_gather(g, wire.contiguous(), self.group)
            _all_gather(gl, wl.contiguous(), self.group)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            ch = GatheredChunk(lo, hi, g, gl, rows, keep, ev, torch.cuda.current_stream(dev))
            chunks.append(ch)
            if consumer is not None:
                consumer(ch)                              # e.g. the CTC head on the gathered chunk, on this range's stream
        else:
            wire = out[lo:hi] if self.wire_dtype is None else out[lo:hi].to(self.wire_dtype)
            g = wire.new_empty((world * n,) + tuple(wire.shape[1:]))
            gl = out_len.new_empty(world * n)
            dist.all_gather_into_tensor(g, wire.contiguous(), group=self.group)
            dist.all_gather_into_tensor(gl, out_len[lo:hi].contiguous(), group=self.group)
            ch = GatheredChunk(lo, hi, g, gl, rows, keep, None, None)
            chunks.append(ch)
            if self.pipelined:                            # CPU tensors (gloo tests of the protocol): same one-call-late consumer
                self._consume_pending((lo, hi))
                self._pending.append((self._call, (lo, hi), ch, consumer))
            elif consumer is not None:
                consumer(ch)

    def _consume_pending(self, key=None):
        """Run the consumers of pending chunks on the CURRENT stream: the one of row range `key`, or (key None) all of them in issue order."""
        keep = []
        for call, k, ch, consumer in self._pending:
            if key is None or k == key:
                ch.wait()
                if consumer is not None:
                    consumer(ch)
            else:
                keep.append((call, k, ch, consumer))
        self._pending = keep

    def flush(self):
        """Pipelined mode: consume what the last call left in flight (on the current stream).  No-op otherwise."""
        self._consume_pending(None)

    # ------------------------------------------------------------------ entry points
    def encode_shard(self, xs: torch.Tensor, ls: torch.Tensor, global_batch: Optional[int] = None, range_pad=None, x_len_host=None,
                     consumer: Optional[Callable] = None) -> Gathered:
        """This rank's rows (already selected; the same number of rows on every rank) -> gathered chunks of the global batch.
        `range_pad`: padded length per row range for `ConformerEncoder.trim_sub_batches` - the SAME list on every rank (the maximum
        over the ranks' shards), so that every rank launches identical shapes and the collectives stay fixed-size.
        `consumer(chunk)`: called right after a range's collective was issued, with that range's stream current (the CTC head of bench.py).
        Ragged encoders (`ConformerEncoder.ragged`): pass `x_len_host`; the row ranges are the rank-independent equal-count cut (set in the
        constructor) or explicit `encoder.sub_batch_bounds` that are IDENTICAL on all ranks - never cuts derived from a rank's own lengths."""
        world = dist.get_world_size(self.group)
        gb = global_batch if global_batch is not None else xs.shape[0] * world
        chunks: List[GatheredChunk] = []
        self._call += 1
        has_cut = hasattr(self.encoder, "ragged_cut")
        saved_cut = self.encoder.ragged_cut if has_cut else None
        try:
            if has_cut:
                self.encoder.ragged_cut = "rows"          # rank-independent ranges for THIS call (explicit sub_batch_bounds still win)
            if self._hooked:
                kw = {"range_pad": range_pad} if range_pad is not None else {}
                if x_len_host is not None:
                    kw["x_len_host"] = x_len_host
                self.encoder(xs, ls, range_hook=lambda lo, hi, out, out_len: self._gather_range(lo, hi, out, out_len, gb, chunks, consumer), **kw)
            else:
                out, out_len = self.encoder(xs, ls)[:2]
                self._gather_range(0, out.shape[0], out, out_len, gb, chunks, consumer)
        finally:
            if has_cut:
                self.encoder.ragged_cut = saved_cut
        if self.pipelined and any(c < self._call for c, _, _, _ in self._pending):
            # row ranges of an EARLIER call that this call did not revisit (the cuts changed): consume them now, on the caller's stream
            stale = [p for p in self._pending if p[0] < self._call]
            self._pending = [p for p in self._pending if p[0] == self._call]
            for _, _, ch, cons in stale:
                ch.wait()
                if cons is not None:
                    cons(ch)
        return Gathered(chunks, gb)

    def __call__(self, x: torch.Tensor, x_len: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Global (B, L) batch (the same tensor on every rank) -> (B, T, D) outputs + lengths in global order."""
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        xs, ls = shard_batch(x, x_len, rank, world, uniform=True)
        g = self.encode_shard(xs, ls, x.shape[0])
        self.flush()                       # a whole-batch call wants THIS call's outputs
        return g.assemble()
