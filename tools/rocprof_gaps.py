#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace db of bench.py: per step (a step starts at a mel_kernel dispatch that follows a
ctc kernel) wall time, busy time (union of the kernel intervals), idle time, dispatches, and the gaps between consecutive dispatches.

    python tools/rocprof_gaps.py run.db [out.txt] ["profiled command"]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else [t for t in tabs if t.startswith("kernels")][0]
    rows = c.execute("select name, start, end from %s order by start" % view).fetchall()
    rows = [(n, s, e) for n, s, e in rows if "rocclr" not in n and "elementwise" not in n]
    steps, cur, seen_ctc = [], [], True
    for n, s, e in rows:
        if "mel_kernel" in n and seen_ctc and cur:
            steps.append(cur); cur = []; seen_ctc = False
        if "ctc_collapse" in n:
            seen_ctc = True
        cur.append((n, s, e))
    if cur:
        steps.append(cur)
    steps = [st for st in steps if any("ctc_collapse" in n for n, _, _ in st)]
    if len(sys.argv) > 3:
        out.write("# %s\n" % sys.argv[3])
    out.write("# per step: wall = last end - first start; busy = union of kernel intervals; idle = wall - busy (us)\n")
    out.write("%5s %10s %10s %10s %10s %8s %12s %12s\n" % ("step", "wall", "busy", "idle", "sum_kernel", "kernels", "median_gap", "gaps>2us"))
    for i, st in enumerate(steps):
        st.sort(key=lambda r: r[1])
        t0, t1 = st[0][1], max(e for _, _, e in st)
        busy, end = 0, t0
        gaps = []
        for n, s, e in st:
            if s > end:
                gaps.append(s - end)
                busy += e - s
            elif e > end:
                busy += e - end
            end = max(end, e)
        gaps.sort()
        med = gaps[len(gaps) // 2] / 1e3 if gaps else 0.0
        out.write("%5d %10.1f %10.1f %10.1f %10.1f %8d %12.2f %12d\n" % (i, (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3,
                                                                     sum(e - s for _, s, e in st) / 1e3, len(st), med, sum(g > 2000 for g in gaps)))


if __name__ == "__main__":
    main()
