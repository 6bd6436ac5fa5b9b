#!/bin/bash
# round 4, call 10: Large's front end writing ragged rows (no pad rows, no gather), tiled fp32 depthwise conv of the label-exact modes,
# kernel-level split GEMM tests; the ragged / wide-config / exact tests, then Large and the two label-exact lines
set -u
out=gpurun_out/r4_10; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -x -k "ragged or shipped or exact or split or Large or sharded" 2>&1 | tail -4
timeout 600 python bench.py --model EfficientConformerCTCLarge --steps 5 --warmup 2 --no-cpu-baseline > $out/large_bench.json 2> $out/large.err
timeout 600 python bench.py --precision split --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_split.json 2>> $out/large.err
timeout 600 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $out/bench_fp32.json 2>> $out/large.err
python - <<PY
import json
for f in ("large_bench", "bench_split", "bench_fp32"):
    d = json.load(open("$out/%s.json" % f))
    print(f, d["value"], d["ms_per_step"], d["check"]["ok"], d["check"]["max_abs_err_vs_oracle"], d["check"].get("argmax_flips_vs_oracle"))
    for k, c in d.get("kernel_classes", {}).items():
        if c["ms_per_step"] > 0: print("   ", k, round(c["ms_per_step"], 3), round(c["frac"], 4))
PY
tail -3 $out/large.err
