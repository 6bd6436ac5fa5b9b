#!/bin/bash
# round 5, call 3: throughput against the batch size / number of row ranges (operating point of the bench)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_03; mkdir -p $out
bench() {
  tag=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['config'].get('row_ranges'))" | tee -a $out/ab.txt
}
bench b256
bench b384 --batch 384
bench b512 --batch 512
bench b512_r4 --batch 512 --ranges 4 --streams 4
bench b512_r6 --batch 512 --ranges 6 --streams 3
bench b768 --batch 768
bench b1024 --batch 1024
bench b1024_r6 --batch 1024 --ranges 6 --streams 3
bench b1024_r2 --batch 1024 --ranges 2 --streams 2
bench b256_r2 --ranges 2 --streams 2
bench b256_r4 --ranges 4 --streams 4
bench b256_r4s3 --ranges 4 --streams 3
bench b256_r6s3 --ranges 6 --streams 3
exit 0
