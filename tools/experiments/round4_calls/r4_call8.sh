#!/bin/bash
# round 4, call 8: PV kernel with key-major transposing stores; bench lines of the two label-exact modes with their roofline legs
set -u
out=gpurun_out/r4_08; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "exact or split or precision or attention_maps or empty_row" 2>&1 | tail -3
timeout 600 python bench.py --precision split --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_split.json 2> $out/bench_split.err
timeout 600 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_fp32.json 2> $out/bench_fp32.err
for f in split fp32; do python - <<PY
import json
d = json.load(open("$out/bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["check"]["ok"], d["check"].get("label_sequences_identical_to_oracle"), d["check"]["max_abs_err_vs_oracle"], d["roofline"]["frac"])
for k, c in d["kernel_classes"].items():
    print("   ", k, round(c["ms_per_step"], 3), round(c["frac"], 4))
PY
done
