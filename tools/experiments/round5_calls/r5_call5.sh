#!/bin/bash
# round 5, call 5: the column-pair chains at the widest stage only; non-temporal hints; 4-wave D = 120 chains
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_05; mkdir -p $out
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench base
bench p4_d193 --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193
bench p4_d193_nt1 --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193 --opt chain_nt=1
bench p4_d193_nt2 --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193 --opt chain_nt=2
bench p4_d193_nt3 --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193 --opt chain_nt=3
bench p3_d193 --opt chain_full_max=256 --opt chain_pair=3 --opt chain_pair_min_d=193
bench p1_d193 --opt chain_full_max=256 --opt chain_pair=1 --opt chain_pair_min_d=193
bench variant1 --opt chain_variant=1
bench p4_d193_variant1 --opt chain_full_max=256 --opt chain_pair=4 --opt chain_pair_min_d=193 --opt chain_variant=1
exit 0
