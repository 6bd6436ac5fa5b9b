#!/usr/bin/env python3
"""ISA guard: no product kernel of libeffconf.so may contain a packed-fp32 VALU instruction with a low-lane operand swizzle.

    python -m efficientconformer_amd._isa_guard [libeffconf.so]   (or tools/check_isa.py)   exit code 1 if a hazardous form is found

Measured on MI355X (profiles/r2_mel_packed_fp32_hazard.txt): v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with an `op_sel` bit set
(a low result lane taking the HIGH half of a 64-bit source pair) return wrong values while a bf16 MFMA of ANOTHER wave executes on
the same SIMD.  libeffconf is therefore compiled with `-target-feature -packed-fp32-ops` (efficientconformer_amd/_build.py); this
script extracts every gfx950 code object from the shared library (clang offload bundles in .hip_fatbin), disassembles it and checks.
Kernels of csrc/debug.hip (victim_kernel / neighbour_kernel) and the diagnostic mel_kernel build with packed fp32 are exempt:
they exist to reproduce the hazard.

Second rule (round 6, profiles/r6_35_side_bisect.txt): the split-precision chain kernels (csrc/sxf_chain.hip: sxc_a_kernel / sxc_b_kernel) must not contain an
exec-masked region at all.  hipcc (ROCm 7.2) places VGPR -> AGPR live-range-split copies inside such regions; a value copied under a narrowed mask and read back after
reconvergence is garbage in the lanes that were off - in those kernels a store address (every ragged forward of one stage faulted).  They are written without a
lane-divergent branch (masked loads from a zero block, masked stores as out-of-range buffer stores); an `s_and_saveexec` / `s_or_saveexec` in their listing means a
source change brought one back."""
import os
import re
import struct
import subprocess
import sys
import tempfile



def _llvm_bin() -> str:
    """llvm-objcopy / llvm-objdump of the ROCm installation hipcc belongs to ($ROCM_PATH, the directory of $HIPCC / hipcc on PATH, /opt/rocm)."""
    import shutil
    cands = []
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "llvm-objcopy")):
            return c
    raise RuntimeError("llvm-objdump / llvm-objcopy not found (looked in %s)" % cands)


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
HAZARD = re.compile(r"v_pk_(add|mul|fma)_f32\b.*\bop_sel:\[[^\]]*1")
PACKED = re.compile(r"v_pk_(add|mul|fma)_f32\b")
EXEMPT = re.compile(r"victim_kernel|neighbour_kernel|mel_pk_build")
EXECMASK = re.compile(r"\bs_(and|or|andn2|xor)_saveexec_b64\b")
BRANCH_FREE = re.compile(r"sxc_[ab]_kernel")


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(_llvm_bin(), "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        blob = open(fat, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off: pos + off + size]
        pos += len(MAGIC)


def scan(lib):
    rows = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            asm = subprocess.run([os.path.join(_llvm_bin(), "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                rows.setdefault(cur, [0, 0, 0])
            elif cur is not None:
                if PACKED.search(line):
                    rows[cur][0] += 1
                    if HAZARD.search(line):
                        rows[cur][1] += 1
                if EXECMASK.search(line):
                    rows[cur][2] += 1
    return rows


def demangle(names):
    import shutil
    if not names:
        return {}
    if shutil.which("c++filt") is None:          # explicit: the exemption list matches DEMANGLED names - without c++filt match the mangled ones
        return {n: n for n in names}
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    if len(out) != len(names):
        return {n: n for n in names}
    return dict(zip(names, out))


MIN_KERNELS = 100      # libeffconf.so holds several hundred kernel instances; a scan that saw (almost) nothing did not scan the library


def check(lib, out=None):
    """Scan `lib`; returns (kernels scanned, product kernels with a hazardous form); `out` (a list) receives the report lines."""
    rows = scan(lib)
    if len(rows) < MIN_KERNELS:
        # fail CLOSED: no gfx950 code object found (a compressed offload bundle, another bundle layout, an empty objdump) would otherwise
        # pass the guard without a single kernel looked at
        raise RuntimeError("ISA guard: only %d gfx950 kernels found in %s (expected >= %d): the offload bundle was not read - "
                           "refusing to pass the library unscanned" % (len(rows), lib, MIN_KERNELS))
    names = demangle(list(rows))
    bad = 0
    for k, (packed, hazard, execmask) in sorted(rows.items()):
        exempt = bool(EXEMPT.search(names[k]))
        if (packed or hazard) and out is not None:
            out.append("%-100s packed-fp32 %5d  op_sel(low) %5d%s" % (names[k][:100], packed, hazard, "  (diagnostic kernel, exempt)" if exempt else ""))
        if hazard and not exempt:
            bad += 1
        if execmask and BRANCH_FREE.search(names[k]):
            if out is not None:
                out.append("%-100s exec-masked regions %d (must be 0: see the module docstring)" % (names[k][:100], execmask))
            bad += 1
    return len(rows), bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeffconf.so")
    lines = []
    n, bad = check(lib, lines)
    print("\n".join(lines))
    print("%d kernels scanned, %d product kernels with hazardous packed-fp32 forms / exec-masked regions in the branch-free kernels" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
