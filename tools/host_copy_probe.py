#!/usr/bin/env python3
"""Host memcpy rates on this box: pageable -> pageable / pinned, torch.copy_ vs numpy, 1 .. N threads (what FrontDoor's packing can reach)."""
import os, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
print("cpus (affinity):", len(os.sched_getaffinity(0)), "torch threads:", torch.get_num_threads())
n = 64 << 20                      # 256 MB of fp32
src = torch.from_numpy(np.random.default_rng(0).standard_normal(n).astype(np.float32))
dst_page = torch.empty(n)
dst_pin = torch.empty(n).pin_memory()
def rate(f, reps=3):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return n * 4 * reps / (time.perf_counter() - t0) / 1e9
print("torch copy_ -> pageable: %.1f GB/s" % rate(lambda: dst_page.copy_(src)))
print("torch copy_ -> pinned:   %.1f GB/s" % rate(lambda: dst_pin.copy_(src)))
a, b = src.numpy(), dst_pin.numpy()
print("numpy copyto -> pinned:  %.1f GB/s" % rate(lambda: np.copyto(b, a)))
torch.set_num_threads(1)
print("torch copy_ -> pinned, 1 intra-op thread: %.1f GB/s" % rate(lambda: dst_pin.copy_(src)))
for w in (2, 4, 8, 16, 32):
    pool = ThreadPoolExecutor(w)
    step = n // w
    def part(i): np.copyto(b[i * step:(i + 1) * step], a[i * step:(i + 1) * step])
    print("numpy copyto, %2d threads -> pinned: %.1f GB/s" % (w, rate(lambda: list(pool.map(part, range(w))))))
    def tpart(i): dst_pin[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step])
    print("torch copy_,  %2d threads -> pinned: %.1f GB/s" % (w, rate(lambda: list(pool.map(tpart, range(w))))))
x = torch.empty(n, device="cuda")
torch.cuda.synchronize()
def h2d(): x.copy_(dst_pin, non_blocking=True); torch.cuda.synchronize()
print("H2D from pinned: %.1f GB/s" % rate(h2d))
