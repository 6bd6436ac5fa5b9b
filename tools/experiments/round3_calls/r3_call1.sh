#!/bin/bash
# round 3, GPU call 1: new tests, full gpu suite, bench with self-check, overlap timeline, extra bench lines, small-batch latency
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c1"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu > "$out/t_round3.log" 2>&1; echo "round3 tests rc=$?" | tee -a "$out/summary.txt"
tail -5 "$out/t_round3.log"
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_round3.py > "$out/t_all.log" 2>&1; echo "all gpu tests rc=$?" | tee -a "$out/summary.txt"
tail -3 "$out/t_all.log"
timeout 600 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
python - <<'PY' | tee -a "$out/summary.txt"
import json
try:
    j=json.load(open("gpurun_out/r3c1/bench.json"))
    print("value %.3fM ms %.3f check %s" % (j["value"]/1e6, j["ms_per_step"], json.dumps(j.get("check"))))
    print("step_ms", j["config"].get("step_ms"), "roofline frac", j["roofline"]["frac"], "whole", j["roofline"]["whole_step"])
except Exception as e: print("bench parse failed", e)
PY
timeout 300 python bench.py --no-trim --no-cpu-baseline --no-roofline > "$out/bench_notrim.json" 2>> "$out/bench.err"
timeout 300 python bench.py --batch 128 --no-cpu-baseline --no-roofline > "$out/bench_b128.json" 2>> "$out/bench.err"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ovl && timeout 600 rocprofv3 --kernel-trace -d /tmp/ovl -o run -- python "$repo/tools/overlap_probe.py" > "$out/overlap_probe.log" 2>&1 )
db=$(find /tmp/ovl -name "*.db" | head -1)
python tools/overlap_timeline.py "$db" "$out/overlap_timeline.txt" 2>&1 | tail -3
head -30 "$out/overlap_timeline.txt"
timeout 300 python tools/small_batch_latency.py > "$out/small_batch_latency.txt" 2>&1; cat "$out/small_batch_latency.txt"
cut -c1-300 "$out/bench_notrim.json" "$out/bench_b128.json"
