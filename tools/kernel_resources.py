#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage -> one line per kernel (VGPR / AGPR / scratch / spills / LDS / occupancy)."""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
    if "error" in line:
        print(line)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print("%-72s V%4d A%4d scratch %4d spill %3d occ %d" % (name, r.get("vgpr", -1), r.get("agpr", -1), r.get("scratch", -1),
                                                          r.get("spill", -1), r.get("occ", -1)))
