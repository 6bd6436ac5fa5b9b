#!/bin/bash
# PMC HBM traffic + SQ counters of the default command on the final tree (split-bf16 CTC head)
set -u
tag=r3_03; repo=$(pwd); out="$repo/gpurun_out/$tag"; mkdir -p "$out"
tools/gpu_pmc.sh "$tag" --steps 5 --warmup 2
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_f && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_f -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_f -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1" > /dev/null
ls -la "$out"; cat "$out/pmc_traffic.json"
