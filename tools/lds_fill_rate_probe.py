#!/usr/bin/env python3
"""L2 -> LDS fill rate of a CU (round 4).  The D = 240 chains stream a 32 KiB weight chunk per barrier interval through every workgroup; their phase profiles
put a refill at ~30 bytes per cycle and CU, LDS-DMA or plain loads alike.  This probe measures the path alone (csrc/debug.hip, lds_fill_kernel): every
workgroup walks the same 1 MiB window (L2 resident, far larger than the 32 KiB L1) in 1-KiB wave-instructions,

    mode 0: global_load_lds_dwordx4 (the chains' refill)          mode 1: global_load_dwordx4, 8 in flight per wave, then ds_write_b128

for 1 / 2 / 4 / 8 waves per workgroup (one workgroup per CU: 128 KiB of LDS each) and 1 / 32 / 256 workgroups (one CU, one XCD's worth, the chip).
Reported: bytes per cycle and workgroup (cycles = s_memtime of wave 0 around the whole walk, median over workgroups).

    python tools/lds_fill_rate_probe.py
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
    window = 1 << 20
    src = torch.randint(0, 255, (window + 4096,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    print("%-28s %-6s %-11s %14s %16s %12s" % ("mode", "waves", "workgroups", "B/cycle/CU", "cycles per 32 KiB", "us (events)"))
    for mode, name in ((0, "LDS-DMA"), (1, "loads + ds_write_b128")):
        for waves in (1, 2, 4, 8):
            for blocks in (1, 32, 256):
                kib = 32 // waves if waves <= 4 else 8            # 32 KiB per pass and workgroup (8 waves: 64 KiB - the multiple-of-8 rule)
                kib = max(8, kib)
                passes = 64
                out = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)

                def go():
                    _lib.check(lib.effconf_debug_lds_fill(mode, blocks, waves, src.data_ptr(), window, kib, passes, out.data_ptr(), st.cuda_stream), "lds_fill")
                go(); go()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); go(); e1.record(st)
                torch.cuda.synchronize()
                o = out.cpu().view(blocks, 2)
                cyc = o[:, 0].float().median().item()
                nbytes = float(o[0, 1])
                print("%-28s %-6d %-11d %14.1f %16.0f %12.1f" % (name, waves, blocks, nbytes / cyc, cyc / (nbytes / 32768), e0.elapsed_time(e1) * 1e3), flush=True)


if __name__ == "__main__":
    main()
