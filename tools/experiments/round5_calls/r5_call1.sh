#!/bin/bash
# round 5, call 1: column-pair chain kernels (chain2.hip) - bit-identity against chain.hip, then A / B of the bench step and per-kernel durations
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_01; mkdir -p $out
for cfgv in "1 192" "2 256"; do set -- $cfgv; timeout 300 python tools/diag_pair_stages.py EfficientConformerCTCSmall 900 $1 $2 2>&1 | tail -12 | tee -a $out/diag.txt; done
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu -x 2>&1 | tail -6 | tee $out/pytest.txt
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
bench pair1 --opt chain_pair=1
bench pair2 --opt chain_pair=2
bench pair2_full256 --opt chain_full_max=256 --opt chain_pair=2
bench pair1_full256 --opt chain_full_max=256 --opt chain_pair=1
bench pair2_full256_1stream --opt chain_full_max=256 --opt chain_pair=2 --streams 1 --ranges 1
bench base_1stream --streams 1 --ranges 1
trace base
trace pair2_full256 --opt chain_full_max=256 --opt chain_pair=2
trace pair1 --opt chain_pair=1
exit 0
