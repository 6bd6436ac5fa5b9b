#!/bin/bash
# round 5, call 4: refills in turns (chain_pair = 3) and the interleaved owner iteration (4): bit-identity, A / B, phase profiles
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_04; mkdir -p $out
for cfgv in "3 192" "4 256" "3 256"; do set -- $cfgv; timeout 300 python tools/diag_pair_stages.py EfficientConformerCTCSmall 900 $1 $2 2>&1 | tail -8 | tee -a $out/diag.txt; done
timeout 300 python tools/diag_pair_stages.py EfficientConformerCTCMedium 700 4 256 2>&1 | tail -8 | tee -a $out/diag.txt
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
trace() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 "$@" > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $*" > /dev/null
  grep "chain" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
bench base
bench pair3 --opt chain_pair=3
bench pair4 --opt chain_pair=4
bench pair3_full256 --opt chain_full_max=256 --opt chain_pair=3
bench pair4_full256 --opt chain_full_max=256 --opt chain_pair=4
bench pair4_full256_min13k --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_m=13000
trace pair4_full256 --opt chain_full_max=256 --opt chain_pair=4
trace pair3 --opt chain_pair=3
for v in "3161 256 4" "2161 256 3"; do set -- $v
  echo "== EFFCONF_CHAIN2_PHASES=$1 chain_full_max=$2 chain_pair=$3" | tee -a $out/phases.txt
  EFFCONF_CHAIN2_PHASES=$1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_full_max=$2 --opt chain_pair=$3 2>&1 | grep "chain2 phases" | tee -a $out/phases.txt
done
exit 0
