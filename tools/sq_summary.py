#!/usr/bin/env python3
"""rocprofv3 --pmc <SQ counters> database -> per-kernel averages (text table for profiles/)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for name, cn, val in c.execute("select name, counter_name, counter_value from pmc_events"):
        a = agg[name][cn]
        a[0] += 1
        a[1] += val
    counters = sorted({cn for k in agg.values() for cn in k})
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc %s ; per-dispatch averages\n# %s\n" % (" ".join(counters), note))
        f.write("%-70s %6s " % ("kernel", "calls") + " ".join("%14s" % cn.replace("SQ_", "")[:14] for cn in counters) + "\n")
        key = counters[0]
        for name in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k][key])[1]):
            n = max(v[0] for v in agg[name].values())
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            f.write("%-70s %6d " % (short, n) + " ".join("%14.4g" % (agg[name][cn][1] / max(agg[name][cn][0], 1)) for cn in counters) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
