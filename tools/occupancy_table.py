#!/usr/bin/env python3
"""Per kernel of a rocprofv3 kernel trace (rocpd sqlite db): workgroup size, LDS bytes, VGPR / AGPR counts and the workgroups / waves a CU can hold
(160 KiB LDS, 512 registers per lane and SIMD, 4 SIMDs) - which resource caps the occupancy of every kernel of the step.

    python tools/occupancy_table.py <db> [out.txt]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else [t for t in tabs if t.startswith("kernels")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    want = [x for x in ("name", "start", "end", "grid_size_x", "grid_size_y", "workgroup_size_x", "lds_block_size", "lds_size", "group_segment_size", "arch_vgpr_count",
                        "accum_vgpr_count", "vgpr_count", "sgpr_count", "scratch_size") if x in cols]
    rows = c.execute("select %s from %s" % (", ".join(want), view)).fetchall()
    agg = {}
    for r in rows:
        d = dict(zip(want, r))
        lds = d.get("lds_block_size", d.get("lds_size", d.get("group_segment_size", 0))) or 0
        key = (d["name"], d.get("workgroup_size_x", 0), lds, d.get("arch_vgpr_count", d.get("vgpr_count", 0)) or 0, d.get("accum_vgpr_count", 0) or 0)
        a = agg.setdefault(key, [0, 0.0, 0])
        a[0] += 1; a[1] += (d["end"] - d["start"]) / 1000.0; a[2] = max(a[2], (d.get("grid_size_x", 0) or 0) * max(1, d.get("grid_size_y", 1) or 1))
    out = ["# columns: " + ", ".join(want), "%-74s %5s %7s %5s %5s | %6s %6s %8s | %6s %10s" % ("kernel", "wg", "lds", "vgpr", "agpr", "wg/CU", "by", "waves/CU", "calls", "total_us")]
    for (name, wg, lds, v, a), (n, tot, grid) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        waves = max(1, wg // 64)
        regs = max(1, v + a)
        per_simd = max(1, 512 // ((regs + 7) // 8 * 8))
        by_regs = per_simd * 4 // waves if waves <= 4 * per_simd else 0
        by_lds = (160 * 1024) // lds if lds else 99
        cap = min(by_regs, by_lds, 32)
        out.append("%-74s %5d %7d %5d %5d | %6d %6s %8d | %6d %10.1f" % (name.replace("(anonymous namespace)::", "")[:74], wg, lds, v, a, cap,
                                                                       "lds" if by_lds < by_regs else "regs", cap * waves, n, tot))
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
