#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c8"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_round3.py -x -q -m gpu > "$out/t.log" 2>&1; echo "tests rc=$?" | tee -a "$out/summary.txt"; tail -4 "$out/t.log"
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-check "$@" > "$out/$tag.json" 2> "$out/$tag.err"; python - "$out/$tag.json" "$tag" <<'PY' | tee -a "$out/summary.txt"
import json,sys
try:
    j=json.load(open(sys.argv[1])); kc=j.get("kernel_classes",{})
    print("%-28s %.2fM %.3fms" % (sys.argv[2], j["value"]/1e6, j["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in kc.items() if v["ms_per_step"]>0})
except Exception as e: print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
run defer1
run defer2
run medium --model EfficientConformerCTCMedium --steps 5 --warmup 2
run large --model EfficientConformerCTCLarge --steps 5 --warmup 2
