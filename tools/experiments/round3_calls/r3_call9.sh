#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c9"; mkdir -p "$out"
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 --streams 1 > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 --streams 1" > /dev/null
cat "$out/sq_counters.txt" | cut -c1-250 | head -40
