#!/bin/bash
# round 4, call 3: bench.py in the two label-exact modes (split: csrc/split.hip; fp32: csrc/exact.hip) and the round-2 workload of the bf16 path
set -u
out=gpurun_out/r4_03; mkdir -p $out
timeout 600 python bench.py --precision split --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_split.json 2> $out/bench_split.err
timeout 600 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_fp32.json 2> $out/bench_fp32.err
timeout 600 python bench.py --ragged 0 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_ragged0.json 2> $out/bench_ragged0.err
for f in split fp32 ragged0; do python - <<PY
import json
d = json.load(open("$out/bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["check"]["ok"], d["check"].get("label_sequences_identical_to_oracle"), d["check"]["max_abs_err_vs_oracle"], d["roofline"]["frac"])
for k, c in d["kernel_classes"].items():
    print("   ", k, round(c["ms_per_step"], 3), round(c["frac"], 4))
PY
done
tail -3 $out/*.err
