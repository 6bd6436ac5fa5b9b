#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB, as reported by rocprofv3 --pmc) -> text summary for profiles/.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced stream
(TCC_EA0_RDREQ tallied at 64 B for 128-B requests) -> the 'fetch_x2' column doubles it; WRITE_SIZE is uncalibrated."""
import json
import sqlite3
import sys
from collections import defaultdict


def load(db, counter):
    c = sqlite3.connect(db)
    agg = defaultdict(lambda: [0, 0.0])
    for name, val in c.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        a = agg[name]
        a[0] += 1
        a[1] += val
    return agg


def main():
    fetch, write, out = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    import re
    st_, wu_ = re.search(r"--steps\s+(\d+)", note), re.search(r"--warmup\s+(\d+)", note)
    nsteps = (int(st_.group(1)) if st_ else 20) + (int(wu_.group(1)) if wu_ else 5)
    tot_f = sum(2 * kb for _, kb in fetch.values()) * 1024.0
    tot_w = sum(kb for _, kb in write.values()) * 1024.0
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), per-dispatch averages in MB\n# %s\n" % note)
        f.write("# forward steps in each pass: %d (steps + warmup; profiled with --no-check --no-roofline: only the timed region's launch shapes)\n" % nsteps)
        f.write("# HBM traffic per step (all kernels, FETCH_SIZE x2 + WRITE_SIZE): %.2f GB  (fetch x2 %.2f GB, write %.2f GB)\n"
                % ((tot_f + tot_w) / nsteps / 1e9, tot_f / nsteps / 1e9, tot_w / nsteps / 1e9))
        f.write("%-100s %7s %12s %12s %12s\n" % ("kernel", "calls", "fetch_MB", "fetch_x2_MB", "write_MB"))
        for name in sorted(fetch, key=lambda k: -fetch[k][1]):
            n, kb = fetch[name]
            wn, wkb = write.get(name, [1, 0.0])
            f.write("%-100s %7d %12.2f %12.2f %12.2f\n" % (name[:100], n, kb / n / 1024, 2 * kb / n / 1024, wkb / max(wn, 1) / 1024))
        # the --pmc passes serialise dispatches: their kernel trace gives every launch's duration WITHOUT a neighbour kernel
        try:
            c = sqlite3.connect(sys.argv[1])
            rows = c.execute("select name, count(*), sum(end - start) / 1000.0 from kernels group by name order by 3 desc").fetchall()
            f.write("#\n# kernel durations in the FETCH_SIZE pass (dispatches serialised by the counter collection), microseconds\n")
            f.write("%-100s %7s %12s %12s\n" % ("kernel", "calls", "total_us", "avg_us"))
            for name, n, tot in rows[:24]:
                f.write("%-100s %7d %12.1f %12.3f\n" % (name[:100], n, tot, tot / n))
        except Exception as e:        # older rocpd schemas: no `kernels` view
            f.write("# (no kernel durations in this database: %s)\n" % e)
    print(open(out).read())
    if len(sys.argv) > 5:          # machine-readable copy for bench.py's roofline.traffic: argv[5] = json path, argv[6..8] = model batch workload, [9] = streams
        kern = {name: {"calls": fetch[name][0], "fetch_x2_bytes": 2 * 1024 * fetch[name][1] / fetch[name][0],
                       "write_bytes": 1024 * write.get(name, [1, 0.0])[1] / max(write.get(name, [1, 0.0])[0], 1)} for name in fetch}
        json.dump({"source": out, "steps_in_profile": nsteps, "hbm_bytes_per_step": (tot_f + tot_w) / nsteps, "model": sys.argv[6], "batch": int(sys.argv[7]), "workload": sys.argv[8],
                   "streams": int(sys.argv[9]) if len(sys.argv) > 9 else 1,
                   "correction": "FETCH_SIZE x2 (gfx950: 128-B read requests tallied as 64 B), WRITE_SIZE as reported; KB -> bytes",
                   "kernels": kern}, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
