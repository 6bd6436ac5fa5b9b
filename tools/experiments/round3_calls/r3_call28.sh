#!/bin/bash
# ordered kernel list of one EfficientConformerCTCLarge step (one stream, one range)
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c28.log; : > $L
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lg
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/lg -o run -- python $repo/bench.py --model EfficientConformerCTCLarge --no-cpu-baseline --no-roofline --no-check --steps 1 --warmup 1 --streams 1 > /tmp/lg.log 2>&1
db=$(find /tmp/lg -name "*.db" | head -1)
grep -o '"ms_per_step": [0-9.]*' /tmp/lg.log >> $L
python $repo/tools/rocprof_dispatches.py "$db" "" | sed 's/void (anonymous namespace):://; s/(anonymous namespace):://g' | cut -c1-110 | tail -420 >> $L
