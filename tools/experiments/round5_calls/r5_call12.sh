#!/bin/bash
# round 5, call 12: chain3 with wave A prefetching the weight stream into L2 (distance in chunks = chain_nt >> 4)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_12; mkdir -p $out
timeout 120 python tools/diag_pair_stages.py EfficientConformerCTCSmall 900 5 256 chain_w2cm=1 chain_nt=96 2>&1 | tail -3 | tee -a $out/diag.txt
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
  grep "chain3" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
bench p4
for d in 2 4 6 10 15; do bench p5_pf$d --opt chain_pair=5 --opt chain_nt=$((d*16)); done
trace p5_pf6 --opt chain_pair=5 --opt chain_nt=96
trace p5_pf15 --opt chain_pair=5 --opt chain_nt=240
echo "== EFFCONF_CHAIN3_PHASES=3 (prefetch 6)" | tee -a $out/phases.txt
EFFCONF_CHAIN3_PHASES=3 timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_pair=5 --opt chain_nt=96 2>&1 | grep "chain3 phases" | tee -a $out/phases.txt
exit 0
