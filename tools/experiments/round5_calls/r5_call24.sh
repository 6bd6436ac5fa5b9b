#!/bin/bash
# round 5, call 24: chain3 with every prologue load issued up front: bit-identity, kernel durations, step
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_24; mkdir -p $out
timeout 120 python tools/diag_pair_stages.py EfficientConformerCTCSmall 900 5 256 2>&1 | tail -2 | tee -a $out/diag.txt
timeout 120 python tools/diag_pair_stages.py EfficientConformerCTCMedium 700 5 256 2>&1 | tail -2 | tee -a $out/diag.txt
bench() {
  tag=$1; shift
  for i in 1 2 3; do timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
trace() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 "$@" > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $*" > /dev/null
  grep "chain3" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
bench p5
trace p5
exit 0
