"""Drop-in for the reference's ``models.encoders.ConformerEncoder`` (reference models/encoders.py:44-142).

Same constructor (``ConformerEncoder(encoder_params: dict)``), same ``forward(x, x_len)`` signature and
return convention, same ``state_dict`` key names (so reference checkpoints load unchanged:
models/model.py:361-384, model_ctc.py:77-88) — but ``forward`` runs the hand-written HIP kernels of
libeffconf.so through the C ABI of include/effconf.h.  The ``nn.Module`` tree below only *holds*
parameters under the reference's names; none of those sub-modules is ever called, and there is no
PyTorch/CPU fallback: without the HIP library or without a GPU tensor, ``forward`` raises.

Differences from the reference, by design (HISTORY.md):
  * eval-mode only (SpecAugment, dropout, variational noise and BatchNorm statistics updates are
    training-time features, reference encoders.py:103-104, layers.py:63);
  * the third return value is a list of ``None`` (the reference returns per-block attention maps that no
    caller consumes: model_ctc.py:63-68, 93; transducer.py:94, 145);
  * weights are packed once (BatchNorm folded, bf16, MFMA-friendly padding); call ``repack()`` after
    mutating parameters in place (``load_state_dict`` / ``.to()`` re-pack automatically).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .config import EncoderPlan, build_plan


class _Holder(nn.Module):
    """Attribute container; exists only so parameter names match the reference."""


def _seq(*mods):
    return nn.Sequential(*[m if m is not None else nn.Identity() for m in mods])


def _ffn_holder(d, dff):
    # reference modules.py:376-383: Sequential(LN, Linear, Swish, Dropout, Linear, Dropout)
    h = _Holder()
    h.layers = _seq(nn.LayerNorm(d, eps=1e-6), nn.Linear(d, dff), None, None, nn.Linear(dff, d), None)
    return h


class _BlockHolder(nn.Module):
    def __init__(self, bp):
        super().__init__()
        d, de = bp.dim_model, bp.dim_expand
        self.feed_forward_module1 = _ffn_holder(d, bp.dim_ffn1)
        att = _Holder()
        att.norm = nn.LayerNorm(d, eps=1e-6)
        mhsa = _Holder()
        for n in ("query_layer", "key_layer", "value_layer", "output_layer", "pos_layer"):
            setattr(mhsa, n, nn.Linear(d, d))
        mhsa.u = nn.Parameter(torch.zeros(d))
        mhsa.v = nn.Parameter(torch.zeros(d))
        att.mhsa = mhsa
        self.multi_head_self_attention_module = att
        conv = _Holder()
        # reference modules.py:498-509: LN, Transpose, Conv1d pw, GLU, Conv1d dw, BN, Swish, Conv1d pw, Transpose, Dropout
        conv.layers = _seq(nn.LayerNorm(d, eps=1e-6), None, nn.Conv1d(d, 2 * de, 1), None,
                           nn.Conv1d(de, de, bp.kernel_size, groups=de), nn.BatchNorm1d(de), None,
                           nn.Conv1d(de, de, 1), None, None)
        self.convolution_module = conv
        self.feed_forward_module2 = _ffn_holder(de, bp.dim_ffn2)
        self.norm = nn.LayerNorm(de, eps=1e-6)
        if bp.transition:   # reference blocks.py:106-110
            self.conv_res = _seq(None, nn.Conv1d(d, de, 1))
        self.stride = bp.conv_stride


class ConformerEncoder(nn.Module):

    def __init__(self, params: dict):
        super().__init__()
        self.params = dict(params)
        self.plan: EncoderPlan = build_plan(params)
        plan = self.plan
        sub = _Holder()
        layers, cin = [], 1
        for c in plan.sub_filters[:plan.sub_layers]:
            layers.append(_seq(nn.Conv2d(cin, c, 3, stride=2, padding=1), nn.BatchNorm2d(c), None))
            cin = c
        sub.layers = nn.ModuleList(layers)
        self.subsampling_module = sub
        self.linear = nn.Linear(plan.dim_in, plan.blocks[0].dim_model)
        self.blocks = nn.ModuleList([_BlockHolder(bp) for bp in plan.blocks])
        self._handle = None
        self._packed = False
        self._packed_device = -1                  # the C library packs weights on the device that is current at pack time
        object.__setattr__(self, "_head", None)   # not a sub-module: keeps state_dict keys equal to the reference's
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._options: Dict[str, int] = {}
        # Sub-batch streams (opt-in here; bench.py runs with 3): `sub_batches = S > 1` runs a forward as S contiguous row ranges on
        # concurrent HIP streams (None = automatic: 2 from `sub_batch_min` utterances on).  Every kernel of the path is a one-round
        # launch that alternates HBM-bound load / store bursts with compute; the other streams' kernels fill one stream's bursts.
        # Un-trimmed ranges share ONE mel launch for the whole batch on the caller's stream (fewer, fuller launches); trimmed ranges
        # run their own mel frontend.  (Round 1 forked at the mel boundary for correctness: its mel kernel returned perturbed spectra
        # next to another stream's MFMA kernels - a packed-fp32 `op_sel` hazard, fixed by building the library without packed-fp32
        # VALU instructions, HISTORY.md section 5a; any number of independent forwards may overlap now.)
        self.sub_batches: Optional[int] = 1
        # at most this many HIP streams for the row ranges (None: one per range).  More ranges than streams = finer length buckets
        # (less padding with trim_sub_batches) at the same concurrency: range i runs on stream i % sub_batch_streams, in order.
        self.sub_batch_streams: Optional[int] = None
        self.sub_batch_min = 64
        self._exact = 0                    # precision: 0 "bf16", 1 "fp32" (csrc/exact.hip), 2 "split" (csrc/split.hip)
        self._exact_packed = 0
        self.stagger_ranges = False        # True: range 0's stream gets the higher priority (set by dist.ShardedEncoder)
        # Trimmed row ranges (opt-in; bench.py's default): with `sub_batches` > 1 every row range runs as ITS OWN batch, padded to its
        # longest utterance instead of the whole batch's - length bucketing inside one forward.  A length-sorted LibriSpeech-shaped
        # batch of 256 spends 29 % of its frames on padding as one batch, 12 % as two ranges.  Pad frames are live in the reference
        # (SURVEY.md 8a), so a trimmed range reproduces the reference run on THAT sub-batch (its rows collated alone), not the run on
        # the whole batch; outputs beyond a range's own T_out are zero-filled.  Needs the lengths on the host (`x_len_host`, or one
        # device sync) or explicit `range_pad` lengths.
        self.trim_sub_batches = False
        # Ragged batches (opt-in; bench.py's default since round 3): every utterance runs at ITS OWN length in one concatenated row space -
        # no pad frames exist, so an utterance's output is the reference's output for that utterance ALONE (batch size 1), whatever else is
        # in the batch (the reference's batched output differs from that by its pad-frame leakage).  With `sub_batches` > 1
        # the row ranges are cut for equal valid frames.  Needs the lengths on the host (`x_len_host`, or one device sync).
        self.ragged = False
        self.sub_batch_bounds = None       # optional row boundaries of the ranges (nsub - 1 increasing indices); default: equal row counts
        # How a RAGGED batch is cut into row ranges when `sub_batch_bounds` is None: "frames" = equal valid frames per range, computed from
        # THIS call's lengths; "rows" = the rank-independent equal-count cut of the rectangular path.  Under `dist.ShardedEncoder` every
        # range ends in a fixed-size collective, so all ranks must cut identical ranges: it sets "rows" (or pass explicit bounds).
        self.ragged_cut = "frames"
        # True (default since round 5): a caller-supplied `x_len_host` is compared with the device lengths - grids and the workspace are sized from
        # the host copy, the kernels index with the device copy (include/effconf.h: effconf_encoder_forward_ragged), so a mismatch is an out-of-bounds
        # access, not a wrong answer.  The comparison costs one synchronisation; it is made once per (device tensor, its version, host values) and
        # remembered, so a serving loop that re-submits the same length tensors (bench.py) pays it on the first call only, and never while a stream
        # is being captured into a graph - and only for an int64 device tensor handed over as is (any conversion makes a temporary whose identity means
        # nothing: those calls compare every time).  False: the caller vouches for `x_len_host[b] == x_len[b]`.
        self.check_host_lengths = True
        self._len_verified = None
        self._sub_streams: Dict[tuple, torch.cuda.Stream] = {}
        self.caller_stream_slot = True     # stream slot 0 of the row ranges = the caller's stream (False: every range on a side stream, as round 2)
        self.eval()

    @staticmethod
    def _capturing(device) -> bool:
        try:
            return device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        except Exception:
            return False

    # ------------------------------------------------------------------ weights
    def attach_head(self, fc: Optional[nn.Linear]):
        """Let the C library also pack the CTC head ``fc`` (reference model_ctc.py:49)."""
        object.__setattr__(self, "_head", fc)
        self._packed = False

    def repack(self):
        self._packed = False

    @property
    def precision(self) -> str:
        """"bf16" (default: bf16 MFMA operands, fp32 accumulation / residual stream), "fp32": the library's exact mode - fp32 operands on
        the fp32 matrix pipe end to end, greedy label sequences identical to the reference's CPU fp32 path wherever its top-2 logit margins
        exceed fp32 summation noise (reference model_ctc.py:99-133), about 10x slower - or "split": the same schedule with every GEMM and the
        attention products on the fp16 matrix pipe with operands split into two fp16 numbers (csrc/split.hip: products accurate to ~2^-21,
        three MFMAs per product): the fast label-exact mode - since round 6 on fused kernels (csrc/sxf.hip: attention without scores in memory) that take
        ragged batches, `causal` configurations and finite contexts like the bf16 path."""
        return ("bf16", "fp32", "split")[self._exact]

    @precision.setter
    def precision(self, value: str):
        if value not in ("bf16", "fp32", "split"):
            raise ValueError("precision must be 'bf16', 'fp32' or 'split'")
        want = ("bf16", "fp32", "split").index(value)
        if want == self._exact:
            return
        self._exact = want
        # the fp32 tensors (and, for "split", the fp16 weight images) are uploaded at finalize: a handle packed for "split" serves all
        # three modes, one packed for "fp32" serves "fp32" and "bf16", one packed for "bf16" only itself
        if self._packed and self._exact_packed >= want:
            _lib.check(_lib.load().effconf_encoder_set_option(self._handle, b"exact_fp32", int(want)), "set_option(exact_fp32)")
        else:
            self._packed = False

    def _param_device(self):
        return self.linear.weight.device

    def set_option(self, name: str, value: int):
        """Forward a tuning / test option to the C library (see effconf_encoder_set_option).  Options are remembered and re-applied
        whenever the weights are packed again (a new handle), before `finalize` - some (chain_full_max) shape what finalize builds."""
        self._options[name] = int(value)
        if name in ("chain_full_max",):
            self._packed = False
        dev = self._param_device()
        if dev.type == "cuda":
            with torch.cuda.device(dev):
                self._ensure_packed()
        else:
            self._ensure_packed()
        _lib.check(_lib.load().effconf_encoder_set_option(self._handle, name.encode(), int(value)), "set_option(%s)" % name)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # real torchaudio registers frontend buffers the reference checkpoints may carry; the native frontend
        # builds its own window / filterbank tables, so accept-and-ignore them.
        sd = {k: v for k, v in state_dict.items() if not k.startswith("preprocessing.")}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._packed = False
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = False
        return r

    def _make_config(self):
        plan = self.plan
        blocks = (_lib.EcBlock * len(plan.blocks))()
        for i, b in enumerate(plan.blocks):
            blocks[i] = _lib.EcBlock(b.dim_model, b.dim_expand, b.dim_ffn1 // b.dim_model, b.num_heads, b.kernel_size,
                                     b.group_size, b.max_pos, b.conv_stride)
        cfg = _lib.EcConfig()
        cfg.n_mels, cfg.sample_rate, cfg.n_fft = plan.n_mels, plan.sample_rate, plan.n_fft
        cfg.win_length, cfg.hop_length = plan.win_length, plan.hop_length
        cfg.normalize, cfg.mean, cfg.std = int(plan.normalize), plan.mean, plan.std
        cfg.sub_layers = plan.sub_layers
        for i in range(4):
            cfg.sub_filters[i] = plan.sub_filters[i] if i < len(plan.sub_filters) else 0
        cfg.num_blocks = len(plan.blocks)
        cfg.blocks = C.cast(blocks, C.POINTER(_lib.EcBlock))
        cfg.vocab_size = self._head.out_features if self._head is not None else 0
        cfg.causal, cfg.left_context, cfg.right_context = int(plan.causal), min(plan.left_context, 1 << 30), min(plan.right_context, 1 << 30)
        return cfg, blocks

    def _ensure_packed(self):
        dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
        if self._packed and self._packed_device == dev:
            return
        lib = _lib.load()
        if self._handle is not None:
            lib.effconf_encoder_destroy(self._handle)
            self._handle = None
        cfg, keep = self._make_config()
        h = lib.effconf_encoder_create(C.byref(cfg))
        if not h:
            raise _lib.EffconfError("effconf_encoder_create: %s" % lib.effconf_last_error().decode())
        self._handle = h
        if self._exact:
            _lib.check(lib.effconf_encoder_set_option(h, b"exact_fp32", int(self._exact)), "set_option(exact_fp32)")
        self._exact_packed = self._exact
        for oname, oval in self._options.items():
            _lib.check(lib.effconf_encoder_set_option(h, oname.encode(), oval), "set_option(%s)" % oname)
        tensors = dict(super().state_dict())
        if self._head is not None:
            tensors["fc.weight"], tensors["fc.bias"] = self._head.weight, self._head.bias
        for key, t in tensors.items():
            if key.endswith("num_batches_tracked"):
                continue
            arr = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(lib.effconf_encoder_load_tensor(h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim),
                       "load_tensor(%s)" % key)
        _lib.check(lib.effconf_encoder_finalize(h), "finalize")
        # this wrapper owns its workspaces exclusively, so the input-independent positional projections may be cached
        _lib.check(lib.effconf_encoder_set_option(h, b"cache_pos_embeddings", 1), "set_option")
        self._ws.clear()
        self._packed = True
        self._packed_device = dev

    def __del__(self):
        try:
            if self._handle is not None and _lib._lib is not None:
                _lib._lib.effconf_encoder_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ forward
    def _workspace(self, batch: int, n: int, from_audio: bool, device, nbytes: Optional[int] = None) -> torch.Tensor:
        # one workspace per (device, stream, entry point), grown to the largest forward seen: forwards enqueued on different streams may
        # overlap on the GPU, forwards on one stream are ordered and share the buffer.  (Keyed by shape, real variable-length traffic
        # went through a fresh allocation - and a reset of every positional-embedding cache - on nearly every forward.)
        key = (str(device), torch.cuda.current_stream(device).cuda_stream, bool(from_audio))
        if nbytes is None:
            nbytes = _lib.load().effconf_encoder_workspace_bytes(self._handle, batch, n, int(from_audio))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            fill = os.environ.get("EFFCONF_POISON_WORKSPACE", "")
            if fill:                # test hook (include/effconf.h): the byte a fresh workspace is filled with - 255 = NaN patterns, 127 =
                ws.fill_(int(fill)) # huge finite values, 0 = zeros; a kernel that reads workspace nobody wrote shows up as a difference
            self._ws[key] = ws
            # a fresh workspace has no valid positional-embedding cache, even if the allocator reuses an old address
            _lib.check(_lib.load().effconf_encoder_set_option(self._handle, b"cache_pos_embeddings", 1), "set_option")
        return ws

    def _run(self, x: torch.Tensor, x_len: Optional[torch.Tensor], from_audio: bool, range_hook: Optional[Callable] = None,
             x_len_host=None, range_pad=None, return_attentions: bool = False):
        if x.is_cuda:
            # the C library allocates packed weights on, and launches on, the CURRENT device: make that the input's device
            with torch.cuda.device(x.device):
                return self._run_on_device(x, x_len, from_audio, range_hook, x_len_host, range_pad, return_attentions)
        return self._run_on_device(x, x_len, from_audio, range_hook, x_len_host, range_pad, return_attentions)

    def _range_pads(self, ranges, n, from_audio, lens, lens_given, x_len_host, range_pad):
        """Padded length (samples / mel frames) of every row range in trimmed mode, or None (every range keeps the batch's)."""
        if not self.trim_sub_batches or len(ranges) < 2 or (range_pad is None and not lens_given):
            return None
        if range_pad is not None:
            pads = [int(v) for v in range_pad]
            if len(pads) != len(ranges):
                raise ValueError("range_pad needs one length per row range (%d)" % len(ranges))
        else:
            hl = x_len_host if x_len_host is not None else lens.cpu()          # without host lengths: one device sync
            hl = [int(v) for v in (hl.tolist() if hasattr(hl, "tolist") else hl)]
            pads = [max(hl[lo:hi]) for lo, hi in ranges]
        floor = self.plan.n_fft // 2 + 1 if from_audio else 1
        pads = [min(n, max(v, floor)) for v in pads]
        return None if all(v == n for v in pads) else pads

    def row_ranges(self, batch: int, nsub: int, host_lens=None):
        """The contiguous row ranges [(lo, hi)] a forward of `batch` utterances runs as (`sub_batches` = nsub): explicit `sub_batch_bounds`,
        else - ragged batches with `ragged_cut == "frames"` - equal VALID frames per range from `host_lens`, else equal utterance counts."""
        if nsub <= 1:
            return [(0, batch)]
        if self.ragged and self.sub_batch_bounds is None and self.ragged_cut != "rows" and host_lens is not None:
            # equal VALID frames per row range (the work of a ragged range), boundaries at multiples of 8 rows
            csum = np.concatenate([[0], np.cumsum(host_lens)])
            cuts = [0]
            for i in range(1, nsub):
                c = int(np.searchsorted(csum, csum[-1] * i / nsub))
                c = max(cuts[-1] + 1, min(batch - (nsub - i), (c + 4) // 8 * 8 if batch >= 16 * nsub else c))
                cuts.append(c)
            cuts.append(batch)
            return [(cuts[i], cuts[i + 1]) for i in range(nsub)]
        if self.sub_batch_bounds is not None:      # explicit row boundaries (e.g. equal padded work per range, staggered ranges)
            cuts = [0] + [int(b) for b in self.sub_batch_bounds] + [batch]
            if len(cuts) != nsub + 1 or any(cuts[i] >= cuts[i + 1] for i in range(nsub)):
                raise ValueError("sub_batch_bounds must be %d increasing row indices inside (0, %d)" % (nsub - 1, batch))
            return [(cuts[i], cuts[i + 1]) for i in range(nsub)]
        # equal utterance counts, boundaries floored to multiples of 16 rows when the ranges are large enough: with a length-sorted
        # batch this moves a few rows from the longest range to the shortest one (B = 256 in 3 ranges: 80 / 80 / 96 rows,
        # +1 % over 85 / 85 / 86 on the LibriSpeech-shaped bench batch; the 4 % first measured for it was the attention kernels'
        # XCD mapping, which only handled B % 8 == 0 - fixed in attention.hip / attention2.hip)
        cuts = [batch * i // nsub for i in range(nsub + 1)]
        if batch >= 32 * nsub:
            cuts = [c - c % 16 for c in cuts[:-1]] + [batch]
        return [(cuts[i], cuts[i + 1]) for i in range(nsub)]

    def _run_on_device(self, x: torch.Tensor, x_len: Optional[torch.Tensor], from_audio: bool, range_hook: Optional[Callable],
                       x_len_host=None, range_pad=None, return_attentions: bool = False):
        if self.training:
            raise RuntimeError("efficientconformer_amd.ConformerEncoder is an inference path: call .eval()")
        if not x.is_cuda:
            raise RuntimeError("ConformerEncoder.forward needs a GPU tensor (HIP path only; no CPU fallback)")
        lib = _lib.load()
        self._ensure_packed()
        x = x.contiguous().float()
        batch, n = (x.shape[0], x.shape[1]) if from_audio else (x.shape[0], x.shape[2])
        lens_given = x_len is not None
        lens = (x_len if lens_given else torch.full((batch,), n, dtype=torch.int64)).to(x.device, torch.int64).contiguous()
        t_out = lib.effconf_encoder_out_frames(self._handle, n, int(from_audio))
        out = torch.empty(batch, t_out, self.plan.dim_out, dtype=torch.float32, device=x.device)
        out_len = torch.empty(batch, dtype=torch.int64, device=x.device)
        fn = lib.effconf_encoder_forward if from_audio else lib.effconf_encoder_forward_mel

        def launch(lo: int, hi: int):      # rows [lo, hi) on the current stream, written straight into out[lo:hi]
            ws = self._workspace(hi - lo, n, from_audio, x.device)
            _lib.check(fn(self._handle, x[lo:].data_ptr(), lens[lo:].data_ptr(), hi - lo, n, out[lo:].data_ptr(), out_len[lo:].data_ptr(),
                          ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream), "encoder_forward")

        nsub = self.sub_batches if self.sub_batches is not None else (2 if batch >= self.sub_batch_min else 1)
        nsub = max(1, min(int(nsub), batch))
        attentions = [None] * len(self.plan.blocks)
        host_lens = None
        if self.ragged:            # the host copy of the lengths ONCE, validated before anything is sized from it (without x_len_host: one device sync)
            if self._exact == 1:
                raise RuntimeError("ragged batches run on the bf16 path and in precision = 'split' (precision = 'fp32' keeps rectangular batches)")
            hl = x_len_host if x_len_host is not None else lens.cpu()
            host_lens = np.ascontiguousarray(np.asarray(hl.cpu() if torch.is_tensor(hl) else hl, dtype=np.int64))
            if host_lens.shape != (batch,):
                raise ValueError("x_len_host needs one length per utterance")
            if x_len_host is not None and self.check_host_lengths and not self._capturing(x.device):
                # Remembered only for the caller's OWN tensor object (no dtype / device conversion happened: `lens is x_len`), held by a strong reference
                # together with its version counter: a converted temporary - or a tensor the caller freed - can come back from the caching allocator at the
                # same address with version 0 and different lengths (round 5 keyed the memo by data_ptr and skipped the comparison then).
                hb = host_lens.tobytes()
                memo = self._len_verified
                same = lens is x_len and memo is not None and memo[0] is x_len and memo[1] == x_len._version and memo[2] == hb
                if not same:
                    if not np.array_equal(host_lens, lens.cpu().numpy()):
                        raise ValueError("x_len_host differs from x_len: the ragged forward sizes its grids and workspace from the host lengths "
                                         "and indexes with the device lengths - they must be the same numbers")
                    self._len_verified = (x_len, x_len._version, hb) if lens is x_len else None
        if return_attentions:
            # the reference's third return value (encoders.py:126-142): one (B, H, Tg, Tg) softmax map per block, written by the library
            # next to the forward (effconf_encoder_set_attention_outputs); the whole batch as ONE rectangular range
            # (ragged batches, round 4: the same rectangles sized for the LONGEST utterance of the batch; an utterance's own Tg x Tg block is
            # its map when run alone, the rest is zero)
            nsub, nb = 1, len(self.plan.blocks)
            heads, tg = (C.c_int32 * nb)(), (C.c_int32 * nb)()
            n_att = int(host_lens.max()) if self.ragged else n
            _lib.check(lib.effconf_encoder_attention_dims(self._handle, n_att, int(from_audio), heads, tg), "attention_dims")
            # zeros: "an utterance's own block = its map, the rest zero" does not depend on the kernel writing every element (opt-in path, cheap)
            attentions = [torch.zeros(batch, heads[k], tg[k], tg[k], dtype=torch.float32, device=x.device) for k in range(nb)]
            _lib.check(lib.effconf_encoder_set_attention_outputs(self._handle, (C.c_void_p * nb)(*[a.data_ptr() for a in attentions]), nb),
                       "set_attention_outputs")
        ranges = self.row_ranges(batch, nsub, host_lens)
        pads = None if self.ragged else self._range_pads(ranges, n, from_audio, lens, lens_given, x_len_host, range_pad)
        fn_ragged = lib.effconf_encoder_forward_ragged

        def launch_ragged(lo: int, hi: int):
            hp = host_lens[lo:hi]
            hptr = hp.ctypes.data_as(C.c_void_p)
            nbytes = lib.effconf_encoder_workspace_bytes_ragged(self._handle, hptr, hi - lo, n, int(from_audio))
            if nbytes == 0:
                raise _lib.EffconfError("ragged batch: a length is out of range (audio: n_fft / 2 < len <= row length)")
            ws = self._workspace(hi - lo, n, from_audio, x.device, nbytes)
            _lib.check(fn_ragged(self._handle, x[lo:].data_ptr(), lens[lo:].data_ptr(), hptr, hi - lo, n, int(from_audio), out[lo:].data_ptr(), t_out,
                                 out_len[lo:].data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream), "encoder_forward_ragged")

        def launch_trimmed(lo: int, hi: int, ni: int):
            # the rows as their own batch, padded to ni <= n: what the reference computes when these utterances are collated alone
            xi = (x[lo:hi, :ni] if from_audio else x[lo:hi, :, :ni]).contiguous()
            ti = lib.effconf_encoder_out_frames(self._handle, ni, int(from_audio))
            oi = out[lo:hi] if ti == t_out else torch.empty(hi - lo, ti, self.plan.dim_out, dtype=torch.float32, device=x.device)
            ws = self._workspace(hi - lo, ni, from_audio, x.device)
            _lib.check(fn(self._handle, xi.data_ptr(), lens[lo:].data_ptr(), hi - lo, ni, oi.data_ptr(), out_len[lo:].data_ptr(),
                          ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream), "encoder_forward")
            if ti != t_out:
                out[lo:hi, :ti] = oi
                out[lo:hi, ti:] = 0

        if nsub == 1:
            try:
                (launch_ragged if self.ragged else launch)(0, batch)
            finally:
                if return_attentions:
                    _lib.check(lib.effconf_encoder_set_attention_outputs(self._handle, None, 0), "set_attention_outputs")
            if range_hook is not None:
                range_hook(0, batch, out, out_len)
        else:
            if from_audio and pads is None and not self.ragged:
                # the whole batch's mel on the caller's stream (see __init__), then forward_mel per row range
                tm = n // self.plan.hop_length + 1
                mel = torch.empty(batch, self.plan.n_mels, tm, dtype=torch.float32, device=x.device)
                _lib.check(lib.effconf_mel_frontend(self._handle, x.data_ptr(), batch, n, mel.data_ptr(),
                                                    torch.cuda.current_stream(x.device).cuda_stream), "mel_frontend")
                x, n, from_audio, fn = mel, tm, False, lib.effconf_encoder_forward_mel
                lens = torch.div(lens, self.plan.hop_length, rounding_mode="floor") + 1
            cur = torch.cuda.current_stream(x.device)
            streams = []
            smax = nsub if not self.sub_batch_streams else max(1, min(int(self.sub_batch_streams), nsub))
            # Stream slot 0 IS the caller's stream; slots 1 .. smax - 1 are side streams that fork from an event recorded before anything of
            # this forward is enqueued.  (Measured on the MI355X: with more than four HIP streams active in the process the ranges stop
            # overlapping - three side streams + the caller's idle stream + ONE more (a collective's stream, a consumer's stream) turned a
            # 5.5 ms step into 7.4 ms, `GPU_MAX_HW_QUEUES` notwithstanding - so the forward itself uses smax streams, not smax + 1, and leaves
            # room for the process group's own stream.)  The side streams are enqueued first: they run while the host enqueues the rest.
            fork = torch.cuda.Event()
            fork.record(cur)
            order = [i for i in range(nsub) if i % smax != 0] + [i for i in range(nsub) if i % smax == 0]
            if self.caller_stream_slot is False:
                order = list(range(nsub))
            for i in order:
                slot = i % smax
                if slot == 0 and self.caller_stream_slot is not False:
                    st = cur
                else:
                    # earlier row ranges get the higher priority (opt-in `stagger_ranges`): range 0 leaves the last stage first
                    prio = -1 if (i == 0 and self.stagger_ranges) else 0
                    key = (str(x.device), slot, prio)        # the priority is part of the key: `stagger_ranges` may change after the first forward
                    if key not in self._sub_streams:
                        self._sub_streams[key] = torch.cuda.Stream(device=x.device, priority=prio)
                    st = self._sub_streams[key]
                    if st not in streams:
                        st.wait_event(fork)              # inputs (and anything queued before this forward) are ready
                        streams.append(st)
                lo, hi = ranges[i]
                with torch.cuda.stream(st):
                    if self.ragged:
                        launch_ragged(lo, hi)
                    elif pads is None:
                        launch(lo, hi)
                    else:
                        launch_trimmed(lo, hi, pads[i])
                    if st is not cur:
                        x.record_stream(st); lens.record_stream(st); out.record_stream(st); out_len.record_stream(st)
                    if range_hook is not None:
                        range_hook(lo, hi, out, out_len)     # called with the range's stream current: rows [lo, hi) of `out` are enqueued
            for st in streams:
                cur.wait_stream(st)                      # joined: the caller continues on its own stream
        return out, (out_len if lens_given else None), attentions

    def forward(self, x: torch.Tensor, x_len: Optional[torch.Tensor] = None, range_hook: Optional[Callable] = None,
                x_len_host=None, range_pad=None, return_attentions: bool = False):
        """x: (B, L) raw 16 kHz audio, x_len: (B,) samples -> (x (B, T_out, D_last), x_len, attentions)
        (reference encoders.py:97-142).

        ``range_hook(lo, hi, out, out_len)`` (optional, not in the reference) is called once per sub-batch row range right after
        that range's kernels were enqueued, with the range's HIP stream current: work enqueued from the hook (an all-gather of
        ``out[lo:hi]``) starts when THAT range is done, while the other ranges are still in their last stage.
        ``x_len_host`` (the lengths as a host sequence) / ``range_pad`` (one padded length per row range) serve ``trim_sub_batches``.
        ``attentions`` is a list of ``None`` per block unless ``return_attentions=True``: then the reference's per-block softmax maps
        (B, H, Tg, Tg) (encoders.py:126-142), at the price of one extra kernel per block and a single row range."""
        return self._run(x, x_len, True, range_hook, x_len_host, range_pad, return_attentions)

    def forward_mel(self, mel: torch.Tensor, mel_len: Optional[torch.Tensor] = None, range_hook: Optional[Callable] = None,
                    x_len_host=None, range_pad=None, return_attentions: bool = False):
        """Enter after AudioPreprocessing: mel (B, n_mels, Tm), lengths in frames (the parity boundary)."""
        return self._run(mel, mel_len, False, range_hook, x_len_host, range_pad, return_attentions)

    def mel_frontend(self, x: torch.Tensor, x_len: Optional[torch.Tensor] = None):
        """AudioPreprocessing.forward (reference modules.py:87-106) on the GPU."""
        lib = _lib.load()
        x = x.contiguous().float()
        tm = x.shape[1] // self.plan.hop_length + 1
        with torch.cuda.device(x.device):
            self._ensure_packed()
            mel = torch.empty(x.shape[0], self.plan.n_mels, tm, dtype=torch.float32, device=x.device)
            _lib.check(lib.effconf_mel_frontend(self._handle, x.data_ptr(), x.shape[0], x.shape[1], mel.data_ptr(),
                                                torch.cuda.current_stream(x.device).cuda_stream), "mel_frontend")
        if x_len is not None:
            x_len = torch.div(x_len, self.plan.hop_length, rounding_mode="floor") + 1
        return mel, x_len

    # ------------------------------------------------------------------ test instrumentation
    def trace_forward_mel(self, mel: torch.Tensor, mel_len: torch.Tensor, arena_bytes: int = 1 << 28):
        """Run forward_mel with the debug trace enabled; returns (out, out_len, {name: tensor})."""
        lib = _lib.load()
        self._ensure_packed()
        arena = torch.zeros(arena_bytes, dtype=torch.uint8, device=mel.device)
        _lib.check(lib.effconf_encoder_set_trace(self._handle, arena.data_ptr(), arena.numel()), "set_trace")
        try:
            out, out_len, _ = self.forward_mel(mel, mel_len)
            torch.cuda.synchronize()
            res = {}
            name = C.create_string_buffer(64)
            off, rows, cols, ld = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            dt = C.c_int32()
            for i in range(lib.effconf_encoder_trace_count(self._handle)):
                _lib.check(lib.effconf_encoder_trace_entry(self._handle, i, name, C.byref(off), C.byref(rows), C.byref(cols),
                                                           C.byref(ld), C.byref(dt)), "trace_entry")
                esz, tdt = (2, torch.bfloat16) if dt.value == 1 else (4, torch.float32 if dt.value == 0 else torch.int32)
                raw = arena[off.value: off.value + rows.value * ld.value * esz]
                t = raw.view(tdt).view(rows.value, ld.value)[:, :cols.value].float().cpu()
                res[name.value.decode()] = t
            return out, out_len, res
        finally:
            lib.effconf_encoder_set_trace(self._handle, None, 0)
