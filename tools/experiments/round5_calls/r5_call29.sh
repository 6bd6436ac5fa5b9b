#!/bin/bash
# round 5, call 29: row ranges / streams of the step re-tuned on the new kernel mix (attention at three workgroups per CU)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_29; mkdir -p $out
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" < /dev/null 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench r3s3
bench r2s2 --ranges 2 --streams 2
bench r4s4 --ranges 4 --streams 4
bench r5s5 --ranges 5 --streams 5
bench r4s3 --ranges 4 --streams 3
bench r6s3 --ranges 6 --streams 3
bench r3s3_balanced --balanced-split
bench r3s3
exit 0
