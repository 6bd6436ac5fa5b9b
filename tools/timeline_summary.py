#!/usr/bin/env python3
"""Where does a step's wall time go BETWEEN the kernels?  Reads the rocpd database of `rocprofv3 --kernel-trace` of the default command
(three row ranges on three HIP streams = three hardware queues) and, for the last steps of the run, reports per queue: kernels, busy time,
idle time between consecutive kernels of the queue (dispatch gaps: a dependent kernel of the same queue starts only after its predecessor
has drained), the time-weighted number of kernels in flight, an estimate of the CUs the kernels in flight ask for, and the fork / join
bubbles of a step (first kernel of the last queue to start, last kernel of the first queue to finish).

    timeline_summary.py <db> <out.txt> "<profiled command>" [steps_to_analyse]
"""
import sqlite3
import sys


def wgs_per_cu(lds, vgpr, agpr, threads):
    waves = max(1, threads // 64)
    regs = max(1, vgpr + agpr)
    per_simd = max(1, 512 // (-(-regs // 8) * 8))          # waves per SIMD by registers
    by_regs = max(1, per_simd * 4 // waves)
    by_lds = max(1, (160 * 1024) // lds) if lds else 16
    return max(1, min(by_regs, by_lds, 32 // waves if waves <= 32 else 1))


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    c = sqlite3.connect(db)
    rows = c.execute("select name, queue_id, start, end, grid_x * grid_y * grid_z, workgroup_x * workgroup_y * workgroup_z, lds_size, vgpr_count, "
                     "accum_vgpr_count from kernels order by start").fetchall()
    rows = [r for r in rows if not r[0].startswith("__amd_rocclr_") and "at::native" not in r[0]]
    mel = [i for i, r in enumerate(rows) if "mel_kernel" in r[0]]
    per_step = 3
    # a step starts at the first of its (three) mel kernels; analyse the last nsteps complete steps
    starts = mel[::per_step]
    if len(starts) < nsteps + 1:
        nsteps = max(1, len(starts) - 1)
    first = starts[-nsteps - 1] if len(starts) > nsteps else starts[0]
    last = starts[-1]
    win = rows[first:last]
    t0, t1 = win[0][2], rows[last][2]
    wall = (t1 - t0) / 1e3
    lines = []
    w = lines.append
    w("# kernel timeline of the last %d steps (rocprofv3 --kernel-trace; times in microseconds)" % nsteps)
    w("# %s" % note)
    w("wall time of the window: %.1f us = %.1f us per step; kernels %d (%.1f per step)" % (wall, wall / nsteps, len(win), len(win) / nsteps))
    queues = sorted(set(r[1] for r in win))
    w("%-8s %8s %12s %12s %12s %14s %14s" % ("queue", "kernels", "busy_us/step", "idle_us/step", "gaps/step", "median_gap_us", "gaps<20us_sum"))
    for q in queues:
        ks = [r for r in win if r[1] == q]
        busy = sum(r[3] - r[2] for r in ks) / 1e3
        gaps = [(b[2] - a[3]) / 1e3 for a, b in zip(ks[:-1], ks[1:])]
        gaps_pos = sorted(g for g in gaps if g > 0)
        med = gaps_pos[len(gaps_pos) // 2] if gaps_pos else 0.0
        small = sum(g for g in gaps_pos if g < 20)
        w("%-8s %8d %12.1f %12.1f %12.1f %14.2f %14.1f" % (q, len(ks), busy / nsteps, sum(gaps_pos) / nsteps, len(gaps_pos) / nsteps, med, small / nsteps))
    # kernels in flight and CU demand over time
    ev = []
    for r in win:
        wgs = max(1, r[4] // max(1, r[5]))
        cus = min(256.0, wgs / wgs_per_cu(r[6], r[7], r[8], r[5]))
        ev.append((r[2], 1, cus)); ev.append((r[3], -1, -cus))
    ev.sort()
    hist = {}
    dem = 0.0; n = 0; prev = ev[0][0]; demand_time = 0.0; capped = 0.0
    for t, dn, dc in ev:
        dt = (t - prev) / 1e3
        if dt > 0:
            hist[n] = hist.get(n, 0.0) + dt
            demand_time += dem * dt
            capped += min(dem, 256.0) * dt
        n += dn; dem += dc; prev = t
    tot = sum(hist.values())
    w("kernels in flight (share of the window): " + "  ".join("%d: %.1f %%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
    w("CUs asked for by the kernels in flight (workgroups / workgroups per CU, capped at 256 per kernel): time average %.0f, capped at the chip %.0f of 256 (%.0f %%)"
      % (demand_time / tot, capped / tot, 100 * capped / tot / 256))
    # per step: fork and join bubbles
    w("%-6s %10s %16s %16s %18s" % ("step", "wall_us", "fork_spread_us", "join_spread_us", "last_queue_alone_us"))
    for si in range(nsteps):
        a = starts[-nsteps - 1 + si]; b = starts[-nsteps + si]
        st = rows[a:b]
        fs = [min(r[2] for r in st if r[1] == q) for q in queues if any(r[1] == q for r in st)]
        le = [max(r[3] for r in st if r[1] == q) for q in queues if any(r[1] == q for r in st)]
        le_sorted = sorted(le)
        w("%-6d %10.1f %16.1f %16.1f %18.1f" % (si, (rows[b][2] - st[0][2]) / 1e3, (max(fs) - min(fs)) / 1e3, (max(le) - min(le)) / 1e3,
                                               (le_sorted[-1] - le_sorted[-2]) / 1e3 if len(le_sorted) > 1 else 0.0))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
