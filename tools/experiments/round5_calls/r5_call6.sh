#!/bin/bash
# round 5, call 6: FFN second weights streamed from chunk-major images (contiguous 1-KiB wave-DMAs)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_06; mkdir -p $out
timeout 300 python tools/diag_pair_stages.py EfficientConformerCTCSmall 900 4 256 chain_w2cm=1 2>&1 | tail -8 | tee -a $out/diag.txt
timeout 300 python tools/diag_pair_stages.py EfficientConformerCTCMedium 700 3 192 chain_w2cm=1 2>&1 | tail -8 | tee -a $out/diag.txt
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
  grep "chain2" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
bench p4 --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193 --opt chain_variant=1
bench p4_cm --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193 --opt chain_variant=1 --opt chain_w2cm=1
bench p4_cm_all --opt chain_full_max=256 --opt chain_pair=4 --opt chain_variant=1 --opt chain_w2cm=1
trace p4_cm_all --opt chain_full_max=256 --opt chain_pair=4 --opt chain_w2cm=1
trace p1_cm_all --opt chain_full_max=256 --opt chain_pair=1 --opt chain_w2cm=1
exit 0
