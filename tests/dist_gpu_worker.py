"""Worker of tests/test_gpu_dist.py: launched twice by torch.distributed.run on ONE GPU (gloo process group, device tensors).
The real HIP encoder behind dist.ShardedEncoder with two sub-batch row ranges: per-range hooks, comm stream, chunk protocol,
record_stream - everything of the multi-GPU path except the RCCL transport - on rectangular ranges AND on the ragged ranges that are
the default of `bench.py --gpus N` (reference sharding: main.py:33-35, 217-220, model_ctc.py:70-75).  Prints DIST_GPU_OK on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientconformer_amd import ModelCTC, named_config, synth  # noqa: E402
from efficientconformer_amd.dist import ShardedEncoder, shard_batch  # noqa: E402


def main():
    backend = os.environ.get("EFFCONF_TEST_BACKEND", "gloo")      # "nccl" = RCCL: one rank per device (a one-GPU box runs it with world size 1)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    name = sys.argv[1] if len(sys.argv) > 1 else "Tiny"
    cfg = named_config(name)
    m = ModelCTC.from_config(cfg)
    sd = synth.make_state_dict(m.encoder.plan, 7, cfg["tokenizer_params"]["vocab_size"], prefix="encoder.")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    enc = m.encoder
    lens = np.array([48000, 41000, 37000, 30000, 22000, 12000, 9000], dtype=np.int64)        # odd batch: shards of 4 and 3 rows
    audio = torch.from_numpy(synth.make_audio(lens, seed=4)).cuda()
    ln = torch.from_numpy(lens).cuda()
    enc.sub_batches = 1
    full, full_len, _ = enc(audio, ln)
    _, labels_full, n_full = m._head(full, full_len)
    enc.sub_batches = 2
    sh = ShardedEncoder(enc)
    ok = True
    for it in range(3 if name == "Tiny" else 2):                                   # repeated: buffers of call k are recycled while call k+1 runs
        out, out_len = sh(audio, ln)
        ok = ok and torch.equal(out, full) and torch.equal(out_len, full_len)
        xs, ls = shard_batch(audio, ln, rank, world, uniform=True)
        g = sh.encode_shard(xs, ls, audio.shape[0])
        seen = {}
        for ch in g.chunks:                               # consumer on the current stream, chunk by chunk (bench.py's head)
            ch.wait()
            _, lab, n = m._head(ch.out, ch.out_len)
            for r in torch.nonzero(ch.keep).flatten().tolist():
                seen[int(ch.rows[r])] = lab[r, :int(n[r])].tolist()
        want = {b: labels_full[b, :int(n_full[b])].tolist() for b in range(audio.shape[0])}
        ok = ok and len(g.chunks) == 2 and seen == want
    # ---- the default of `bench.py --gpus N`: RAGGED row ranges (every utterance at its own length), one fixed-size collective per range
    #      issued from the range's stream, the CTC head as the consumer of every gathered chunk.  A ragged forward is batch-invariant, so
    #      the gathered outputs must equal the UNSHARDED ragged run bit for bit (fp32 wire), or its bf16 rounding (bf16 wire); labels from
    #      the gathered chunks equal the head run locally on the same values.  Cuts: ShardedEncoder's rank-independent equal-count cut,
    #      then explicit (pinned) bounds as bench.py sets them.
    if enc.precision == "bf16":
        enc.ragged, enc.sub_batches, enc.sub_batch_bounds = True, 1, None
        rag, rag_len, _ = enc(audio, ln, x_len_host=lens)
        ok = ok and torch.equal(rag_len, full_len)
        xs, ls = shard_batch(audio, ln, rank, world, uniform=True)
        idx = list(range(rank, audio.shape[0], world))
        hl = lens[idx + [idx[-1]] * (xs.shape[0] - len(idx))]
        enc.check_host_lengths = True
        for wire in (None, torch.bfloat16):
            want_out = rag if wire is None else rag.to(wire)
            _, lab_w, n_w = m._head(want_out.float(), rag_len)
            for bounds in (None, [1], [3]):
                enc.sub_batches, enc.sub_batch_bounds = 2, bounds
                shr = ShardedEncoder(enc, wire_dtype=wire)
                ok = ok and enc.ragged_cut == "frames"     # the shared encoder keeps ITS policy; "rows" is asserted per encode_shard call (and dropping the previous wrapper cannot reset it)
                seen, outs = {}, {}

                def consumer(ch):                          # runs with the range's stream current, right behind the collective
                    _, lab, n = m._head(ch.out.float(), ch.out_len)
                    for r in torch.nonzero(ch.keep).flatten().tolist():
                        seen[int(ch.rows[r])] = (lab[r], n[r])
                        outs[int(ch.rows[r])] = (ch.out[r], ch.out_len[r])
                g = shr.encode_shard(xs, ls, audio.shape[0], x_len_host=hl, consumer=consumer)
                torch.cuda.synchronize()
                ok = ok and len(g.chunks) == 2 and sorted(seen) == list(range(audio.shape[0]))
                for b in range(audio.shape[0]):
                    o, l = outs[b]
                    nl = int(seen[b][1])
                    ok = ok and torch.equal(o, want_out[b]) and int(l) == int(rag_len[b]) and nl == int(n_w[b]) and \
                        torch.equal(seen[b][0][:nl], lab_w[b, :nl])
        # ---- pipelined protocol (ShardedEncoder(pipelined=True), bench.py's default at N > 1): a range's collective is issued asynchronously
        #      (RCCL: async_op on the process group's stream) and its consumer runs one call later on that range's stream; flush() drains.
        #      Two different batches in flight, consumers must see exactly their own call's gathered rows.
        enc.sub_batches, enc.sub_batch_bounds = 2, [2]
        lens2 = np.array([44000, 40000, 33000, 31000, 20000, 15000, 8000], dtype=np.int64)
        audio2 = torch.from_numpy(synth.make_audio(lens2, seed=9)).cuda()
        ln2 = torch.from_numpy(lens2).cuda()
        if audio2.shape[1] < audio.shape[1]:
            audio2 = torch.nn.functional.pad(audio2, (0, audio.shape[1] - audio2.shape[1]))
        enc.sub_batches = 1
        rag2, rag2_len, _ = enc(audio2, ln2, x_len_host=lens2)
        enc.sub_batches = 2
        shp = ShardedEncoder(enc, pipelined=True)
        got = []
        for tag, (au, l_d, l_h, ref, ref_len) in (("a", (audio, ln, lens, rag, rag_len)), ("b", (audio2, ln2, lens2, rag2, rag2_len)), ("a2", (audio, ln, lens, rag, rag_len))):
            xs_, ls_ = shard_batch(au, l_d, rank, world, uniform=True)
            idx_ = list(range(rank, au.shape[0], world))
            hl_ = l_h[idx_ + [idx_[-1]] * (xs_.shape[0] - len(idx_))]
            shp.encode_shard(xs_, ls_, au.shape[0], x_len_host=hl_,
                             consumer=lambda ch, tag=tag, ref=ref, ref_len=ref_len: got.append((tag, ch.lo, ch.out.clone(), ch.out_len.clone(), ch.rows, ch.keep, ref, ref_len)))
        n_before_flush = len(got)
        shp.flush()
        torch.cuda.synchronize()
        ok = ok and n_before_flush == 4 and [g_[0] for g_ in got] == ["a", "a", "b", "b", "a2", "a2"]
        for tag, lo, o, ol, rows, keep, ref, ref_len in got:
            for r in torch.nonzero(keep).flatten().tolist():
                b = int(rows[r])
                ok = ok and torch.equal(o[r], ref[b]) and int(ol[r]) == int(ref_len[b])
        enc.ragged, enc.sub_batches, enc.sub_batch_bounds, enc.check_host_lengths = False, 1, None, False
    torch.cuda.synchronize()
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_GPU_OK" if float(flag) == 1.0 else "DIST_GPU_FAILED")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
