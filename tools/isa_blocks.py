#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_blocks.py file.s 'relpos_attention2_kernelILi64ELi4ELi1' [--dump LABEL]

Prints, for every basic block that holds at least one MFMA (or every block with --all): VALU / transcendental / MFMA / LDS /
VMEM / SALU / s_nop / s_waitcnt counts.  The loop a kernel lives in is the block set between a label and the s_cbranch back to it.
"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
show_all = "--all" in sys.argv
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or "; -- End function" in lines[i])
TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_log_", "v_sqrt_", "v_sin_", "v_cos_")
blocks, cur = [], {"label": "entry", "ins": []}
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\w+):", s)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "ins": []}
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur["ins"].append(s.split(";")[0].strip())
blocks.append(cur)


def classify(i):
    op = i.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op == "s_nop": return "nop"
    if op == "s_waitcnt": return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


print("%-14s %5s %5s %5s %5s %5s %5s %5s %5s %4s  branch-targets" % ("block", "valu", "trans", "mfma", "lds", "vmem", "salu", "nop", "wait", "bar"))
for b in blocks:
    cnt = {}
    for i in b["ins"]:
        c = classify(i)
        cnt[c] = cnt.get(c, 0) + 1
    if b["label"] == dump:
        print("\n".join(b["ins"]))
    if not show_all and not cnt.get("mfma") and len(b["ins"]) < 40:
        continue
    tg = [i.split()[-1] for i in b["ins"] if i.startswith(("s_cbranch", "s_branch"))]
    print("%-14s %5d %5d %5d %5d %5d %5d %5d %5d %4d  %s" % (b["label"], cnt.get("valu", 0), cnt.get("trans", 0), cnt.get("mfma", 0), cnt.get("lds", 0),
                                                          cnt.get("vmem", 0), cnt.get("salu", 0), cnt.get("nop", 0), cnt.get("wait", 0), cnt.get("barrier", 0), " ".join(tg)))
