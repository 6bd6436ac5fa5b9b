#!/bin/bash
# round 4, call 28: kernel timeline of the default step (gaps between kernels of a queue, kernels in flight, fork / join bubbles)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_28; mkdir -p $out
cmd="python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && rocprofv3 --kernel-trace -d /tmp/tl -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 > "$out/trace.log" 2>&1 )
db=$(find /tmp/tl -name "*.db" | head -1)
python tools/timeline_summary.py "$db" "$out/timeline.txt" "$cmd" 4
grep '^{' $out/trace.log | cut -c1-200
# attention stage 0 (head width 96): one staging set (default) against two (option attn_waves 2 restores them)
for i in 1 2; do for o in "" "--opt attn_waves=2"; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check $o 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$o', d['value'], d['ms_per_step'])" | tee -a $out/att96_sets.txt; done; done
