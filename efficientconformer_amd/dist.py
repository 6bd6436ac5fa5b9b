"""Multi-GPU: utterances shard data-parallel, one process per GPU, RCCL all-gather of encoder outputs.

The reference's only parallelism is data parallelism (DDP + DistributedSampler, reference main.py:33-35,
model_ctc.py:70-75, functions.py:167-170); in eval mode its encoder performs no collective at all and only
decoded strings are gathered (model.py:463-466).  The path shards naturally along the batch: utterance b's
output depends only on its own row and on the padded length (pad frames are live), so

  * rank r takes rows r::world of the length-sorted global batch (round-robin keeps shards length-balanced);
  * every shard keeps the GLOBAL padded length, so results are bit-equal to the unsharded run;
  * one all-gather per forward of the (B_loc, T_out, D) outputs (+ lengths) — `backend="nccl"` is RCCL over
    xGMI on ROCm — issued on a side stream so it overlaps the next batch's encoder kernels;
  * weights are replicated (<= 251 MB bf16), there is no other data-path collective.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(batch: int, rank: int, world: int) -> torch.Tensor:
    return torch.arange(rank, batch, world) if rank < batch else torch.empty(0, dtype=torch.int64)


def shard_batch(x: torch.Tensor, x_len: torch.Tensor, rank: int, world: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rows rank::world of a (B, L) batch; L (the global pad length) is preserved on purpose."""
    idx = shard_rows(x.shape[0], rank, world).to(x.device)
    return x.index_select(0, idx).contiguous(), x_len.index_select(0, idx).contiguous()


def all_gather_outputs(out: torch.Tensor, out_len: torch.Tensor, global_batch: int, group=None,
                       wire_dtype: Optional[torch.dtype] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather (B_loc, T, D) shard outputs into the global (B, T, D) order of ``shard_rows``.

    Shards may differ by one row when world does not divide B; they are padded to ceil(B/world) rows for the
    collective (a single fixed-size all_gather_into_tensor — the RCCL fast path) and trimmed afterwards."""
    world = dist.get_world_size(group)
    per = (global_batch + world - 1) // world
    wire = out if wire_dtype is None else out.to(wire_dtype)
    pad = per - wire.shape[0]
    if pad:
        wire = torch.cat([wire, wire.new_zeros((pad,) + tuple(wire.shape[1:]))], 0)
        out_len = torch.cat([out_len, out_len.new_zeros(pad)], 0)
    gathered = wire.new_empty((world * per,) + tuple(wire.shape[1:]))
    glen = out_len.new_empty(world * per)
    dist.all_gather_into_tensor(gathered, wire.contiguous(), group=group)
    dist.all_gather_into_tensor(glen, out_len.contiguous(), group=group)
    # gathered row (r*per + i) is global row i*world + r
    order = torch.arange(world * per, device=out.device).view(world, per).t().reshape(-1)[:global_batch]
    return gathered.index_select(0, order), glen.index_select(0, order)


class ShardedEncoder:
    """encoder(x, x_len) on this rank's shard + all-gather on a side stream (when CUDA), global order restored."""

    def __init__(self, encoder: Callable, group=None, wire_dtype: Optional[torch.dtype] = None):
        self.encoder, self.group, self.wire_dtype = encoder, group, wire_dtype
        self._side = None

    def __call__(self, x: torch.Tensor, x_len: torch.Tensor):
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        xs, ls = shard_batch(x, x_len, rank, world)
        if xs.is_cuda and self._side is not None:
            # the previous call's all-gather is complete before this call's mel kernel starts (normally it finished long ago under
            # the caller's head): mel launches are kept away from other streams' kernels they were not swept against
            torch.cuda.current_stream(xs.device).wait_stream(self._side)
        out, out_len = self.encoder(xs, ls)[:2]
        if out.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=out.device)
            self._side.wait_stream(torch.cuda.current_stream(out.device))
            with torch.cuda.stream(self._side):
                res = all_gather_outputs(out, out_len, x.shape[0], self.group, self.wire_dtype)
            self.pending = self._side          # caller: torch.cuda.current_stream().wait_stream(enc.pending) before use
            return res
        return all_gather_outputs(out, out_len, x.shape[0], self.group, self.wire_dtype)
