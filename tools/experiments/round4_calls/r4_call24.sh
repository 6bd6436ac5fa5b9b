#!/bin/bash
# round 4, call 24: EfficientConformerCTCLarge (BASELINE configs[2] per GPU): PMC HBM traffic and SQ counters of the final tree
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_24; mkdir -p $out
tools/gpu_pmc.sh r4_24 --model EfficientConformerCTCLarge --steps 3 --warmup 1
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_l && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_l -o run -- python "$repo/bench.py" --model EfficientConformerCTCLarge --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_l -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python bench.py --model EfficientConformerCTCLarge --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1" > /dev/null
head -12 $out/pmc_hbm_traffic.txt | cut -c1-150; head -8 $out/sq_counters.txt | cut -c1-200
