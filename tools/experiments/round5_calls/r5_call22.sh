#!/bin/bash
# round 5, call 22: wide models - the 256 x 256 LDS-DMA GEMM below 200 tiles (three streams: a launch costs its CU time, not its latency)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_22; mkdir -p $out
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 5 --warmup 2 "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
for m in EfficientConformerCTCLarge EfficientConformerCTCMedium ConformerCTCLarge; do
  bench ${m}_200 --model $m
  for t in 140 100 64 32; do bench ${m}_$t --model $m --opt wide_gemm=$t; done
done
exit 0
