#!/usr/bin/env python3
"""Time the tiled GEMM kernels alone on the layer shapes of the wide configurations (GPU box only).

    python tools/gemm_bench.py [--rows 9600,19200,38400]

For every (M, N, K, epilogue) it launches effconf_debug_gemm with each tile choice (wide = 1: gemm.hip 128 x 128 register staged;
2 / 3: gemm256.hip 256 x 256 / 256 x 128 LDS-DMA) 20 times between two events and prints TFLOP/s; `auto` is launch_gemm's own pick.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from efficientconformer_amd import _lib  # noqa: E402

EPI = {"f32": 0, "bf16": 1, "swish": 2, "resid": 3, "glu": 4}


def ru(x, m):
    return (x + m - 1) // m * m


def run(M, N, K, epi, wide, reps=20):
    lib = _lib.load_debug()
    a = torch.randn(M, ru(K, 8), device="cuda").to(torch.bfloat16)
    w = (torch.randn(ru(N, 128), ru(K, 64), device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.zeros(ru(N, 128), device="cuda")
    e = EPI[epi]
    if e in (0, 3):
        c = torch.empty(M, N, device="cuda"); ldc = N
    elif e == 4:
        c = torch.empty(M, ru(N // 2, 8), device="cuda", dtype=torch.bfloat16); ldc = ru(N // 2, 8)
    else:
        c = torch.empty(M, ru(N, 8), device="cuda", dtype=torch.bfloat16); ldc = ru(N, 8)
    r = torch.randn(M, N, device="cuda") if e == 3 else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    call = lambda: lib.effconf_debug_gemm(p(a), a.shape[1], p(w), w.shape[1], p(b), M, N, K, e, wide, p(c), ldc, p(r), N, C.c_float(0.5),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if call() != 0:
        return None
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * M * N * K / ms / 1e9, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="4800,9600,19200,38400")
    a = ap.parse_args()
    layers = []
    for D in (512, 720):
        layers += [("ffn_a", 4 * D, D, "swish"), ("ffn_b", D, 4 * D, "resid"), ("qkv", 3 * D, D, "bf16"), ("out", D, D, "resid"),
                   ("pw1", 64 * ((D + 31) // 32), D, "glu"), ("pw2", D, D, "resid")]
    print("%-8s %6s %6s %6s %-6s | %21s %21s %21s %21s" % ("layer", "M", "N", "K", "epi", "128x128 TF (ms)", "256x256 TF (ms)", "256x128 TF (ms)", "auto TF (ms)"))
    for M in [int(x) for x in a.rows.split(",")]:
        for name, N, K, epi in layers:
            cells = []
            for wide in (1, 2, 3, 0):
                r = run(M, N, K, epi, wide)
                cells.append("%9.1f (%7.4f)" % r if r else "%21s" % "n/a")
            print("%-8s %6d %6d %6d %-6s | %s" % (name, M, N, K, epi, " ".join("%21s" % c for c in cells)), flush=True)


if __name__ == "__main__":
    main()
