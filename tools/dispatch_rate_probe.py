#!/usr/bin/env python3
"""How fast does the MI355X start workgroups?  (round 4: the attention kernel's duration did not react to anything done INSIDE its workgroups -
8 % shorter workgroup life, three instead of two per CU - so the question is whether ~1000 workgroups x 256 threads are dispatch-bound.)

neighbour_kernel<4> (csrc/debug.hip: 256 threads, `iters` dependent mul + add pairs on 16 values per lane, one 64-byte store per lane at the
end) for workgroup counts 256 .. 8192, dynamic LDS 0 / 54 KiB (three per CU) / 76 KiB (two per CU) and two amounts of work per workgroup:
if the time of a launch grows with the workgroup count while the chip is far from full, the launch rate is the limit.

    python tools/dispatch_rate_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
from efficientconformer_amd import _lib       # noqa: E402


def main():
    lib = _lib.load_debug()
    dev = torch.device("cuda", 0)
    buf = torch.zeros(8192 * 256 * 16, device=dev)
    st = torch.cuda.current_stream(dev)
    print("%-8s %-10s %-8s %10s %14s %16s" % ("lds_KiB", "workgroups", "iters", "us/launch", "ns/workgroup", "workgroups/us"))
    for lds in (0, 54272, 77824):
        for iters in (1, 400):
            for blocks in (256, 512, 1024, 2048, 4096, 8192):
                def go(n):
                    for _ in range(n):
                        _lib.check(lib.effconf_debug_neighbour(4, blocks, lds, iters, buf.data_ptr(), buf.numel(), st.cuda_stream), "neighbour")
                go(3)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); go(20); e1.record(st)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                print("%-8.0f %-10d %-8d %10.2f %14.1f %16.1f" % (lds / 1024, blocks, iters, us, us * 1e3 / blocks, blocks / us), flush=True)


if __name__ == "__main__":
    main()
