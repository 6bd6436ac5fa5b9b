"""Build libeffconf.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m efficientconformer_amd._build [--force]

The shared object is written next to this file so that it travels to the GPU box with the source
snapshot; it is git-ignored.  hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libeffconf.so")
LIB_DEBUG = os.path.join(HERE, "libeffconf_debug.so")
SOURCES = ["gemm.hip", "gemm256.hip", "rsgemm.hip", "chain.hip", "chain2.hip", "chain3.hip", "norm.hip", "conv.hip", "sublinear.hip", "sublinear2.hip", "sublinear3.hip", "conv2.hip", "mel.hip", "ctc.hip", "rnnt.hip", "attention.hip", "attention2.hip", "exact.hip", "split.hip", "sxf.hip", "sxf_ffn.hip", "sxf_chain.hip", "sxf_sub.hip", "hostpack.hip", "encoder.hip"]
# (source, object, extra flags): further compilations of a source under other flags
# No packed-fp32 VALU instructions in product kernels: v_pk_{add,mul,fma}_f32 with an op_sel low-lane swizzle return wrong values
# next to another wave's bf16 MFMA on gfx950 (measured: profiles/r2_mel_packed_fp32_hazard.txt; guard: _isa_guard.py).
# Cost of the flag for the whole library: 7.56 -> 7.60 ms per bench step.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# libeffconf_debug.so (tests / tools only, include/effconf_debug.h) = the product objects with encoder.hip and mel.hip recompiled under -DEFFCONF_DEBUG_ABI
# (the effconf_debug_* entry points, the diagnostic mel_kernel variants) + debug.hip and the packed-fp32 build of mel.hip - the hazard reproducers need the
# instructions they demonstrate, so those two are compiled WITHOUT the flag below.  Nothing of this is linked into libeffconf.so.
DEBUG_REPLACES = {"encoder.hip": "encoder_dbg.o", "mel.hip": "mel_dbg.o", "sxf_ffn.hip": "sxf_ffn_dbg.o"}
DEBUG_OBJECTS = [("debug.hip", "debug.o", []), ("mel.hip", "mel_pk.o", ["-DMEL_PK_BUILD"])]
# sxf_chain.hip: the source-scheduled F2 + Swish body is ~400 unrolled iterations of a 13-way switch - beyond the default cost bound of `#pragma unroll`, and a loop
# left rolled indexes its register arrays dynamically (= scratch memory)
PER_SOURCE = {"sxf_chain.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"], "sxf_sub.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"], "sublinear3.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}
# -fvisibility=hidden: only what include/effconf.h / effconf_debug.h declare (under `#pragma GCC visibility push(default)`) is a dynamic symbol
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result",
         "-Wno-inline-asm"]   # rowstat.h clobbers m0 on purpose (LDS-DMA destination register)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(LIB_DEBUG):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_DEBUG))       # a failed link of the diagnostic library must not leave a stale one behind
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", h) for h in ("effconf.h", "effconf_debug.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "effconf.h"),
                                                                                       os.path.join(HERE, "..", "include", "effconf_debug.h"),
                                                                                       os.path.abspath(__file__)]
    hdr_t = max(os.path.getmtime(h) for h in headers if os.path.exists(h))

    def compile_one(job):
        src, objname, extra = job
        obj = os.path.join(objdir, objname)
        # incremental: an object newer than its source, every header and this script (the flags) is reused unless --force
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            return obj
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj
    jobs = [(s, s.replace(".hip", ".o"), NO_PACKED_FP32 + PER_SOURCE.get(s, [])) for s in SOURCES]
    djobs = [(s, o, NO_PACKED_FP32 + ["-DEFFCONF_DEBUG_ABI"]) for s, o in DEBUG_REPLACES.items()] + DEBUG_OBJECTS
    with ThreadPoolExecutor(max_workers=min(8, len(jobs) + len(djobs))) as ex:
        allobjs = list(ex.map(compile_one, jobs + djobs))
    objs, dobjs = allobjs[:len(jobs)], allobjs[len(jobs):]
    replaced = {os.path.join(objdir, s.replace(".hip", ".o")) for s in DEBUG_REPLACES}
    debug_link = [o for o in objs if o not in replaced] + dobjs
    # link to a temporary name, run the ISA guard on it, and only then move it into place: a library that fails the guard never
    # becomes importable (before: the first build raised AFTER writing libeffconf.so and the next import passed silently)
    tmp = LIB + ".tmp"
    vscript = ["-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + vscript + ["-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    try:
        check_isa(tmp)
    except Exception:
        os.remove(tmp)
        raise
    os.replace(tmp, LIB)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + vscript + ["-o", LIB_DEBUG] + debug_link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link of the diagnostic library failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built %s (%d KB) and %s (%d KB)" % (LIB, os.path.getsize(LIB) // 1024, os.path.basename(LIB_DEBUG), os.path.getsize(LIB_DEBUG) // 1024))
    return LIB


def check_isa(lib: str = LIB) -> None:
    """Fail the build if a product kernel carries a hazardous packed-fp32 form (efficientconformer_amd/_isa_guard.py)."""
    from . import _isa_guard
    lines = []
    n, bad = _isa_guard.check(lib, lines)
    if bad:
        raise RuntimeError("ISA guard failed: %d product kernels with hazardous packed-fp32 forms\n%s" % (bad, "\n".join(lines)[-3000:]))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
