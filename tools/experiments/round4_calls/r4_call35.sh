#!/bin/bash
# round 4, call 35: A / B on ONE box - attention2.hip of this round (lookup in one round trip, prologue loads together, branch-free block loads) against
# round 3's file (kept for the run as tools/experiments/ab/attention2_round3.hip = git show f15a03e:efficientconformer_amd/csrc/attention2.hip; removed afterwards), rebuilt in place on the box between the runs
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_35; mkdir -p $out
run() {
  tag=$1
  for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt; done
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1" > /dev/null
  grep "attention2" $out/kernel_stats_$tag.txt | cut -c1-60,110-200 | tee -a $out/ab.txt
}
run new3wg
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --opt attn_waves=2 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('new2wg', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt; done
cp tools/experiments/ab/attention2_round3.hip efficientconformer_amd/csrc/attention2.hip
python -m efficientconformer_amd._build 2>&1 | tail -1
run round3
