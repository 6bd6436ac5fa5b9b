#!/usr/bin/env python3
"""What is the "more than four active HIP streams stop overlapping" cliff (HISTORY.md section 6)?

N streams each run a chain of K small kernels that fill 1 / 8 of the chip (32 workgroups of a VALU loop): if the streams overlap, the wall
time stays flat up to N = 8; where they serialise it grows by one chain per extra stream.  Run for N = 1 .. 8, in child processes with
GPU_MAX_HW_QUEUES = default / 2 / 4 / 8 (the variable is read when the HIP runtime initialises, so it has to be in the child's
environment before `import torch`), and with the streams created with and without priorities.

    python tools/stream_cliff_probe.py            # the sweep (spawns children)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(prio):
    import torch
    sys.path.insert(0, ROOT)
    from efficientconformer_amd import _lib
    lib = _lib.load_debug()
    dev = torch.device("cuda", 0)
    buf = torch.zeros(1 << 20, device=dev)
    K, blocks, iters = 40, 32, 40           # ~30 us per kernel (a dependent exp / rcp / fma chain), 32 workgroups = 1 / 8 of the CUs

    def chain(st):
        for _ in range(K):
            _lib.check(lib.effconf_debug_neighbour(1, blocks, 0, iters, buf.data_ptr(), buf.numel(), st.cuda_stream), "neighbour")
    out = []
    for n in range(1, 9):
        streams = [torch.cuda.Stream(device=dev, priority=(-1 if (prio and i == 0) else 0)) for i in range(n)]
        for st in streams:
            chain(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream(dev)
        e0.record(cur)
        for st in streams:
            st.wait_event(e0)
        for rep in range(3):
            for st in streams:
                chain(st)
        for st in streams:
            cur.wait_stream(st)
        e1.record(cur)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 3)
    print("GPU_MAX_HW_QUEUES=%-8s priorities=%d  ms per round of N chains, N = 1..8: %s   (x chain time: %s)"
          % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), prio, " ".join("%6.2f" % v for v in out), " ".join("%4.2f" % (v / out[0]) for v in out)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
        sys.exit(0)
    for q in (None, "2", "4", "8", "16"):
        for prio in (0, 1):
            env = dict(os.environ)
            env.pop("GPU_MAX_HW_QUEUES", None)
            if q:
                env["GPU_MAX_HW_QUEUES"] = q
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(prio)], env=env, capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("GPU_MAX")]
            print(lines[0] if lines else ("FAILED: " + r.stderr[-400:]))
