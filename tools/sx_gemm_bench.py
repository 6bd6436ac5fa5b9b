#!/usr/bin/env python3
"""Split-precision GEMM kernel alone (csrc/split.hip: sx_gemm_kernel) on the layer shapes of the label-exact step: accuracy against a float64
product and the achieved rate.  `TF/s` counts the algorithmic 2 M N K; the kernel issues three fp16 MFMAs per product, so its ceiling is a
third of the 2.5 PF dense fp16 peak.

    python tools/sx_gemm_bench.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientconformer_amd import _lib  # noqa: E402


def split_images(w):
    """h = fp16(w), l = fp16((w - h) * 2048), columns zero padded to a multiple of 32 and stored k-tile major (what effconf_encoder_finalize builds)."""
    n, k = w.shape
    ldh = (k + 31) // 32 * 32
    h = w.astype(np.float16)
    l = ((w - h.astype(np.float32)) * 2048.0).astype(np.float16)
    hi = np.zeros((n, ldh), np.float16); lo = np.zeros((n, ldh), np.float16)
    hi[:, :k] = h; lo[:, :k] = l
    # k-tile major: [ldh / 32][n][32]
    tile = lambda a: np.ascontiguousarray(a.reshape(n, ldh // 32, 32).transpose(1, 0, 2))
    return tile(hi), tile(lo), ldh


def run(m, n, k, epi, reps=20, check=True):
    lib = _lib.load_debug()
    g = np.random.default_rng(m + n + k)
    a = g.standard_normal((m, k), dtype=np.float32)
    w = (g.standard_normal((n, k), dtype=np.float32) / np.sqrt(k)).astype(np.float32)
    bias = (0.1 * g.standard_normal(n)).astype(np.float32)
    r = g.standard_normal((m, n), dtype=np.float32)
    hi, lo, ldh = split_images(w)
    ad, hd, ld, bd, rd = (torch.from_numpy(x).cuda() for x in (a, hi.view(np.int16), lo.view(np.int16), bias, r))
    c = torch.empty(m, n, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(lib.effconf_debug_sx_gemm(ad.data_ptr(), k, hd.data_ptr(), ld.data_ptr(), ldh, bd.data_ptr(), m, n, k, epi, c.data_ptr(), n,
                                             rd.data_ptr(), n, C.c_float(0.5), st), "sx_gemm")
    call()
    torch.cuda.synchronize()
    err = None
    if check:
        rows = np.linspace(0, m - 1, min(m, 256)).astype(np.int64)
        ref = a[rows].astype(np.float64) @ w.astype(np.float64).T + bias
        if epi == 1:
            ref = ref / (1.0 + np.exp(-ref))
        elif epi == 2:
            ref = r[rows] + 0.5 * ref
        err = float(np.abs(c.cpu().numpy()[rows] - ref).max() / max(np.abs(ref).max(), 1.0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * m * n * k / us / 1e6, err


if __name__ == "__main__":
    print("%8s %6s %6s %4s %10s %10s %12s" % ("M", "N", "K", "epi", "us", "TF/s", "rel err"))
    for (m, n, k, epi) in [(48000, 480, 120, 1), (48000, 120, 480, 2), (48000, 360, 120, 0), (48000, 120, 120, 2), (48000, 240, 120, 0),
                           (145000, 480, 120, 1), (145000, 120, 480, 2), (72000, 672, 168, 1), (72000, 168, 672, 2), (36000, 960, 240, 1),
                           (36000, 240, 960, 2), (50000, 120, 4800, 0), (1600, 120, 120, 0), (35000, 2880, 720, 1), (35000, 720, 2880, 2),
                           (1000, 100, 100, 0), (333, 77 * 4, 52, 1)]:
        us, tf, err = run(m, n, k, epi)
        print("%8d %6d %6d %4d %10.1f %10.1f %12.2e" % (m, n, k, epi, us, tf, err))
