#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats) rocpd sqlite database into a small text summary for profiles/."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n# %s\n" % note)
        f.write("%-110s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            f.write("%-110s %8d %14.1f %12.3f %8.2f\n" % (name[:110], calls, tot, avg, pct))
        f.write("# sum of kernel time: %.1f us\n" % sum(r[2] for r in rows))
    print(open(out).read())


if __name__ == "__main__":
    main()
