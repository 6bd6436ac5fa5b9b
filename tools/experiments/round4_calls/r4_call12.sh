#!/bin/bash
# round 4, call 12: Large's front end on ragged rows (conv writes the ragged rows, Linear on valid rows, no gather): tests, line, kernel trace
set -u
out=gpurun_out/r4_12; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -x -k "ragged or shipped or Large or sharded" 2>&1 | tail -4
timeout 600 python bench.py --model EfficientConformerCTCLarge --steps 5 --warmup 2 --no-cpu-baseline > $out/large_bench.json 2> $out/large.err
python - <<PY
import json
d = json.load(open("$out/large_bench.json"))
print("large", d["value"], d["ms_per_step"], d["check"]["ok"], d["check"]["max_abs_err_vs_oracle"], d["check"].get("argmax_flips_vs_oracle"))
for k, c in d.get("kernel_classes", {}).items():
    if c["ms_per_step"] > 0: print("   ", k, round(c["ms_per_step"], 3), round(c["frac"], 4))
PY
tools/gpu_profile.sh r4_12_large --model EfficientConformerCTCLarge --steps 3 --warmup 1
head -14 gpurun_out/r4_12_large/kernel_stats.txt | cut -c1-64,110-190; grep -n "subsample_conv\|gather_rows\|256, 0" gpurun_out/r4_12_large/kernel_stats.txt | cut -c1-64,110-190
