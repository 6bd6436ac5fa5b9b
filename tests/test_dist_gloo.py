"""world_size-2 gloo test (CPU): sharded == unsharded, bit for bit, including an odd batch and ragged lengths.
The per-shard encoder here is the oracle (tests may use it); on GPUs the same helpers wrap the HIP encoder."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from efficientconformer_amd import named_config, synth
from efficientconformer_amd.config import build_plan
from efficientconformer_amd.dist import ShardedEncoder, shard_batch, shard_rows


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ref_encoder as R
    plan = build_plan(named_config("Tiny")["encoder_params"])
    sd = synth.make_state_dict(plan, 7)
    mel, lens = synth.make_mel(5, 80, 64, [64, 60, 41, 33, 17], seed=99)      # odd batch: shards of 3 and 2 rows
    mel, lens = torch.from_numpy(mel), torch.from_numpy(lens)

    def enc(m, l):
        with torch.no_grad():
            return R.encoder_from_mel(m, l, sd, plan)
    full, full_len = enc(mel, lens)
    out, out_len = ShardedEncoder(enc)(mel, lens)
    ok = torch.equal(out, full) and torch.equal(out_len, full_len)

    class Ranged:      # the ConformerEncoder protocol: forward(x, x_len, range_hook) calls back per sub-batch row range
        def forward(self, m, l, range_hook=None):
            o, ol = enc(m, l)
            cut = (m.shape[0] + 1) // 2
            for lo, hi in ((0, cut), (cut, m.shape[0])):
                if hi > lo:
                    range_hook(lo, hi, o, ol)
            return o, ol, None
        __call__ = forward
    sh = ShardedEncoder(Ranged())
    xs, ls = shard_batch(mel, lens, rank, world, uniform=True)
    g = sh.encode_shard(xs, ls, mel.shape[0])
    out2, out_len2 = g.assemble()
    ok = ok and len(g.chunks) == 2 and torch.equal(out2, full) and torch.equal(out_len2, full_len)
    # chunk rows map to global rows: a consumer working chunk by chunk (the CTC head in bench.py) sees every utterance once
    seen = sorted(int(r) for c in g.chunks for r in c.rows[c.keep])
    ok = ok and seen == list(range(mel.shape[0]))
    for c in g.chunks:
        ok = ok and torch.equal(c.out[c.keep], full[c.rows[c.keep]])
    out3, _ = ShardedEncoder(enc, wire_dtype=torch.bfloat16)(mel, lens)
    ok = ok and out3.dtype == torch.bfloat16 and torch.equal(out3, full.to(torch.bfloat16))
    # pipelined protocol (dist.py): a range's consumer runs ONE CALL LATER (its collective overlaps the next call's encoder); flush() drains
    shp = ShardedEncoder(Ranged(), pipelined=True)
    got = []
    mel_b, lens_b = synth.make_mel(5, 80, 64, [64, 50, 44, 30, 21], seed=5)
    mel_b, lens_b = torch.from_numpy(mel_b), torch.from_numpy(lens_b)
    full_b, _ = enc(mel_b, lens_b)
    xb, lb = shard_batch(mel_b, lens_b, rank, world, uniform=True)
    shp.encode_shard(xs, ls, mel.shape[0], consumer=lambda c: got.append(("a", c)))
    ok = ok and got == []                                       # nothing consumed yet: call 1's collectives are "in flight"
    shp.encode_shard(xb, lb, mel.shape[0], consumer=lambda c: got.append(("b", c)))
    ok = ok and [t for t, _ in got] == ["a", "a"]               # call 2 consumed call 1's two chunks, range by range
    shp.flush()
    ok = ok and [t for t, _ in got] == ["a", "a", "b", "b"]
    for tag, c in got:
        ref = full if tag == "a" else full_b
        ok = ok and torch.equal(c.out[c.keep], ref[c.rows[c.keep]])
    o4, l4 = ShardedEncoder(Ranged(), pipelined=True)(mel, lens)     # whole-batch call: flushes itself
    ok = ok and torch.equal(o4, full) and torch.equal(l4, full_len)
    # the wrapped encoder's ragged cut policy: "rows" DURING every encode_shard call - whatever wrappers of the same encoder were
    # created or dropped before - and the encoder's own policy at all other times
    import gc
    class Spying(Ranged):
        ragged_cut, seen_cut = "frames", []

        def forward(self, m, l, range_hook=None):
            self.seen_cut.append(self.ragged_cut)
            return Ranged.forward(self, m, l, range_hook)
        __call__ = forward
    shared = Spying()
    first = ShardedEncoder(shared)
    second = ShardedEncoder(shared, pipelined=True)
    del first
    gc.collect()                                                # round 5: the dropped wrapper's __del__ wrote "frames" back under the live one
    ok = ok and shared.ragged_cut == "frames"
    second.encode_shard(xs, ls, mel.shape[0])
    second.close()
    third = ShardedEncoder(shared)
    third.encode_shard(xs, ls, mel.shape[0])
    ok = ok and shared.seen_cut == ["rows", "rows"] and shared.ragged_cut == "frames"
    q.put((rank, bool(ok), float((out - full).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res


def test_shard_rows_cover_batch():
    for b, w in ((5, 2), (8, 8), (7, 4), (3, 8)):
        rows = sorted(int(i) for r in range(w) for i in shard_rows(b, r, w))
        assert rows == list(range(b))
