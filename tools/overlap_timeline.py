#!/usr/bin/env python3
"""rocprofv3 --kernel-trace db of tools/overlap_probe.py -> does the comm-stream traffic of row range i run UNDER the encoder kernels of the
other ranges?  The comm stream is the one whose dispatches are all copy kernels; its dispatches are grouped into collectives (gaps < 25 us);
for every collective: start / duration, the part of it during which encoder kernels (chain / attention / dwconv / subsampling / mel) of other
streams execute, and the top kernels it runs under.

    python tools/overlap_timeline.py run.db [out.txt]
"""
import sqlite3
import sys


def is_copy(n):
    return "copyBuffer" in n or "direct_copy" in n or ("elementwise_kernel" in n and "opy" in n)


def is_encoder(n):
    return any(k in n for k in ("chain_kernel", "chain2_kernel", "chain3_kernel", "relpos_attention", "dwconv_kernel", "dwconv_mfma_kernel", "sublinear", "mel_kernel", "rs_gemm", "gemm_kernel", "ffn_fused"))


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    rows = c.execute("select s.kernel_name, d.start, d.end, d.stream_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                     "order by d.start").fetchall()
    by_stream = {}
    for n, s, e, st in rows:
        by_stream.setdefault(st, []).append((n, s, e))
    # comm stream: only copy kernels, the most of them (weight uploads sit on the default stream next to everything else)
    first_enc = min((s for n, s, e, st in rows if is_encoder(n)), default=0)
    cands = [(sum(1 for n, s, e in v if s > first_enc), st) for st, v in by_stream.items() if all(is_copy(n) for n, _, _ in v)]   # (uploads precede the first encoder kernel)
    for st in list(by_stream):
        by_stream[st] = [(n, s, e) for n, s, e in by_stream[st] if s > first_enc]
    if not cands:
        out.write("no copy-only stream found; streams: %s\n" % {st: len(v) for st, v in by_stream.items()})
        return
    comm = max(cands)[1]
    encs = [(n, s, e, st) for n, s, e, st in rows if is_encoder(n) and st != comm]
    groups, g = [], []
    for n, s, e in by_stream[comm]:
        if g and s - g[-1][2] > 25000:
            groups.append(g); g = []
        g.append((n, s, e))
    if g:
        groups.append(g)
    groups = [g for g in groups if len(g) >= 4]                 # a collective = world - 1 chunk copies (+ casts)
    skip = len(groups) // 4                                     # warm-up steps
    out.write("# tools/overlap_probe.py under rocprofv3 --kernel-trace (one MI355X): the per-range all-gather of dist.ShardedEncoder replaced by D2D copies of\n")
    out.write("# the SAME BYTES an 8-rank all-gather delivers to this rank, on the comm stream (stream %s: %d copy dispatches, %d collectives).\n" % (comm, len(by_stream[comm]), len(groups)))
    out.write("# covered = part of a collective's duration during which encoder kernels of the row ranges' streams execute.\n")
    out.write("%4s %12s %10s %10s %6s  %s\n" % ("#", "start_ms", "dur_us", "covered_us", "%", "runs under (kernel: overlap us)"))
    t0 = groups[skip][0][1] if len(groups) > skip else 0
    tot_cov = tot_dur = 0
    ei = 0
    for gi, g in enumerate(groups[skip:]):
        a, b = g[0][1], max(e for _, _, e in g)
        ivs, names = [], {}
        for n, s, e, st in encs:
            if e <= a or s >= b:
                continue
            ivs.append((max(a, s), min(b, e)))
            k = n.split("(")[0].replace("void (anonymous namespace)::", "")[:44]
            names[k] = names.get(k, 0) + min(b, e) - max(a, s)
        ivs.sort()
        cov, end = 0, a
        for s_, e_ in ivs:
            if e_ > end:
                cov += e_ - max(s_, end); end = e_
        top = ", ".join("%s: %.0f" % (k, v / 1e3) for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:3])
        if gi < 18:
            out.write("%4d %12.3f %10.0f %10.0f %6.0f  %s\n" % (gi, (a - t0) / 1e6, (b - a) / 1e3, cov / 1e3, 100.0 * cov / max(b - a, 1), top or "-"))
        tot_cov += cov; tot_dur += b - a
    if tot_dur:
        out.write("# %d collectives after warm-up: %.0f %% of their time runs under encoder kernels of other streams (%.1f of %.1f ms)\n"
                  % (len(groups) - skip, 100.0 * tot_cov / tot_dur, tot_cov / 1e6, tot_dur / 1e6))


if __name__ == "__main__":
    main()
