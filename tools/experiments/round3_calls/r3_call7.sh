#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c7"; mkdir -p "$out"
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-check "$@" > "$out/$tag.json" 2> "$out/$tag.err"; python - "$out/$tag.json" "$tag" <<'PY' | tee -a "$out/summary.txt"
import json,sys
try:
    j=json.load(open(sys.argv[1])); kc=j.get("kernel_classes",{})
    print("%-28s %.2fM %.3fms" % (sys.argv[2], j["value"]/1e6, j["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in kc.items() if v["ms_per_step"]>0})
except Exception as e: print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
run base
run att96_one_set --opt attn_waves=1
run base2
run att96_one_set2 --opt attn_waves=1
tools/gpu_pmc.sh r3c7pmc --steps 5 --warmup 2
tail -25 gpurun_out/r3c7pmc/pmc_hbm_traffic.txt
