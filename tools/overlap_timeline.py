#!/usr/bin/env python3
"""rocprofv3 --kernel-trace db of tools/overlap_probe.py -> does the comm-stream traffic of range i run UNDER the encoder kernels of the later
ranges?  For every step and every stand-in collective (a run of copy kernels between two ctc kernels): start / end relative to the step,
the encoder kernels (chain / attention / dwconv / ...) that execute during it, and how much of its duration is covered by them.

    python tools/overlap_timeline.py run.db [out.txt]
"""
import sqlite3
import sys


def is_copy(n):
    return "copyBuffer" in n or ("elementwise_kernel" in n and "copy" in n.lower()) or "direct_copy" in n


def is_encoder(n):
    return any(k in n for k in ("chain_kernel", "relpos_attention", "dwconv_kernel", "sublinear", "mel_kernel", "rs_gemm", "gemm_kernel", "ffn_fused"))


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else [t for t in tabs if t.startswith("kernels")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    qcol = next((q for q in ("stream_id", "queue_id", "queue") if q in cols), None)
    rows = c.execute("select name, start, end%s from %s order by start" % ((", " + qcol) if qcol else "", view)).fetchall()
    rows = [(r[0], r[1], r[2], r[3] if qcol else 0) for r in rows]
    # steps: split at the first mel_kernel after a ctc_collapse
    steps, cur, seen = [], [], True
    for r in rows:
        if "mel_kernel" in r[0] and seen and cur:
            steps.append(cur); cur = []; seen = False
        if "ctc_collapse" in r[0]:
            seen = True
        cur.append(r)
    steps.append(cur)
    steps = [st for st in steps if sum("ctc_collapse" in r[0] for r in st) >= 3 and any(is_copy(r[0]) for r in st)][2:]   # skip warm-up
    out.write("# tools/overlap_probe.py under rocprofv3 --kernel-trace: stand-in collectives (7 D2D copies of a range's encoder output = the bytes an\n")
    out.write("# 8-rank all-gather delivers to this rank) on the comm stream vs the encoder kernels of the other row ranges.  Times in us from the step's\n")
    out.write("# first kernel.  covered = part of the collective's duration during which at least one encoder kernel of ANOTHER stream executes.\n")
    tot_cov = tot_dur = 0.0
    for si, st in enumerate(steps):
        t0 = st[0][1]
        wall = (max(r[2] for r in st) - t0) / 1e3
        copies = [r for r in st if is_copy(r[0]) and (r[2] - r[1]) > 3000]      # > 3 us: the chunk copies, not the small trims
        # group the copies into collectives: consecutive copies on the comm queue
        groups, g = [], []
        for r in copies:
            if g and r[1] - g[-1][2] > 150000:
                groups.append(g); g = []
            g.append(r)
        if g:
            groups.append(g)
        encs = [r for r in st if is_encoder(r[0])]
        last_enc_end = max(r[2] for r in encs)
        out.write("step %d: wall %.0f us, %d encoder kernels, last encoder kernel ends at %.0f us\n" % (si, wall, len(encs), (last_enc_end - t0) / 1e3))
        for gi, g in enumerate(groups):
            a, b = g[0][1], max(r[2] for r in g)
            ivs = sorted((max(a, r[1]), min(b, r[2])) for r in encs if r[2] > a and r[1] < b)
            cov, end = 0, a
            for s_, e_ in ivs:
                if e_ > end:
                    cov += e_ - max(s_, end); end = e_
            names = {}
            for r in encs:
                ov = min(b, r[2]) - max(a, r[1])
                if ov > 0:
                    k = r[0].split("(")[0].replace("void (anonymous namespace)::", "")[:40]
                    names[k] = names.get(k, 0) + ov
            top = ", ".join("%s %.0f us" % (k, v / 1e3) for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:3])
            out.write("   collective %d: %2d copies, %7.0f .. %7.0f us (%.0f us), covered %.0f us = %3.0f %%   under: %s\n"
                      % (gi, len(g), (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, cov / 1e3, 100.0 * cov / max(b - a, 1), top or "-"))
            tot_cov += cov; tot_dur += b - a
    if tot_dur:
        out.write("# all steps: %.0f %% of the collectives' time runs under encoder kernels\n" % (100.0 * tot_cov / tot_dur))


if __name__ == "__main__":
    main()
