#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c6"; mkdir -p "$out"
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-check "$@" > "$out/$tag.json" 2> "$out/$tag.err"; python - "$out/$tag.json" "$tag" <<'PY' | tee -a "$out/summary.txt"
import json,sys
try:
    j=json.load(open(sys.argv[1])); kc=j.get("kernel_classes",{})
    print("%-28s %.2fM %.3fms" % (sys.argv[2], j["value"]/1e6, j["ms_per_step"]), {k: round(v["ms_per_step"],2) for k,v in kc.items() if v["ms_per_step"]>0})
except Exception as e: print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
run base
run tiled240_s3 --opt chain_max_dim=192 --opt wide_gemm=2 --opt tiled_min_k=200
run tiled240_s1 --opt chain_max_dim=192 --opt wide_gemm=2 --opt tiled_min_k=200 --streams 1
run tiled240_s2 --opt chain_max_dim=192 --opt wide_gemm=2 --opt tiled_min_k=200 --streams 2
run rs240_s3 --opt chain_max_dim=192
run tiled168_s2 --opt chain_max_dim=128 --opt wide_gemm=2 --opt tiled_min_k=160 --streams 2
run base_s1 --streams 1
