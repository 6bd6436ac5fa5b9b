#!/bin/bash
# round 5, call 11: chain3 with the refill in front of the second GEMM (two iterations of prefetch distance) against behind it
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_11; mkdir -p $out
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
trace() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 "$@" > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $*" > /dev/null
  grep "chain3\|chain2" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
bench p4
bench p5_first --opt chain_pair=5
bench p5_behind --opt chain_pair=5 --opt chain_nt=4
trace p5_first --opt chain_pair=5
for k in 3 1; do
  echo "== EFFCONF_CHAIN3_PHASES=$k (refill first)" | tee -a $out/phases.txt
  EFFCONF_CHAIN3_PHASES=$k timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_pair=5 2>&1 | grep "chain3 phases" | tee -a $out/phases.txt
done
exit 0
