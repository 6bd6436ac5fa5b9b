#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats) rocpd sqlite database into a small text summary for profiles/.

    rocprof_summary.py <db> <out.txt> "<profiled command>"

The command string is parsed for --steps / --warmup: the profile holds steps + warmup forward steps (profile with --no-check --no-roofline,
so that no other launch shape - the self-check's single-utterance forwards, the roofline leg's serial passes - dilutes the per-kernel
averages), and the per-step column divides by that count."""
import re
import sqlite3
import sys


def steps_in(cmd):
    st = re.search(r"--steps\s+(\d+)", cmd)
    wu = re.search(r"--warmup\s+(\d+)", cmd)
    return (int(st.group(1)) if st else 20) + (int(wu.group(1)) if wu else 5)


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    n = steps_in(note)
    clean = "--no-check" in note and "--no-roofline" in note
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n# %s\n" % note)
        f.write("# forward steps in this profile: %d (steps + warmup)%s\n" % (n, "" if clean else
                "  - NOT a clean profile: without --no-check --no-roofline other launch shapes are mixed into the averages"))
        f.write("%-110s %8s %14s %12s %8s %12s %10s\n" % ("kernel", "calls", "total_us", "avg_us", "pct", "us_per_step", "calls/step"))
        for name, calls, tot, avg, pct in rows:
            f.write("%-110s %8d %14.1f %12.3f %8.2f %12.1f %10.1f\n" % (name[:110], calls, tot, avg, pct, tot / n, calls / n))
        f.write("# sum of kernel time: %.1f us = %.1f us per step\n" % (sum(r[2] for r in rows), sum(r[2] for r in rows) / n))
        setup = [r for r in rows if r[0].startswith("__amd_rocclr_") or "at::native" in r[0]]
        if setup:
            t = sum(r[2] for r in setup)
            f.write("# of which runtime / torch kernels outside the forward step (parameter upload at pack time, input synthesis, output checks - no hipMemcpy "
                    "or torch op runs inside effconf_encoder_forward): %.1f us; library kernels only: %.1f us per step\n" % (t, (sum(r[2] for r in rows) - t) / n))
    print(open(out).read())


if __name__ == "__main__":
    main()
