#!/usr/bin/env python3
"""Per-dispatch durations (us) of kernels matching a substring, in launch order, from a rocprofv3 rocpd sqlite db."""
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("kernels") or t == "kernels"]
    view = "kernels" if "kernels" in tabs else kd[0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    rows = c.execute("select name, start, end, grid_size_x, workgroup_size_x from %s order by start" % view).fetchall() if "grid_size_x" in cols else \
        [(r[0], r[1], r[2], 0, 0) for r in c.execute("select name, start, end from %s order by start" % view)]
    for name, st, en, g, w in rows:
        if pat in name:
            print("%9.2f us  grid %7d wg %4d  %s" % ((en - st) / 1000.0, g, w, name[:90]))


if __name__ == "__main__":
    main()
