#!/bin/bash
# round 4, call 9: evidence on the tree of the split-mode commit: whole -m gpu suite, smoke, default bench (roofline on the timed launch
# shape + full-batch shape, cpu_baseline), clean kernel stats / PMC traffic / SQ counters of the default command, the label-exact lines
set -u
tag=r4_09; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $out/pytest_gpu.txt
timeout 900 python bench.py --steps 50 --warmup 10 > $out/bench.json 2> $out/bench.err
tools/gpu_profile.sh $tag --steps 5 --warmup 2
tools/gpu_pmc.sh $tag --steps 5 --warmup 2
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
repo=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_f && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_f -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_f -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1" > /dev/null
timeout 600 python bench.py --precision split --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_split.json 2>> $out/bench.err
timeout 600 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_fp32.json 2>> $out/bench.err
cat $out/pytest_gpu.txt; python - <<PY
import json
for f in ("bench", "bench_split", "bench_fp32"):
    d = json.load(open("$out/%s.json" % f))
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["check"]["ok"], "frac", r["frac"], "traffic", r["traffic"], "alg", r["alg_bytes_per_launch"], "whole", r["whole_step"]["mfma_frac"], r.get("full_batch_launch", {}).get("frac"))
PY
head -30 $out/kernel_stats.txt | cut -c1-70,110-190; head -8 $out/pmc_hbm_traffic.txt
