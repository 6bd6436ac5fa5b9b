#!/bin/bash
# round 5, call 25: wide models - every K > tiled_min_k layer on LayerNorm + tiled 256 x 256 GEMMs (wide_gemm = 2) against the row-stationary fused kernels
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_25; mkdir -p $out
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 5 --warmup 2 "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
for m in EfficientConformerCTCLarge EfficientConformerCTCMedium; do
  bench ${m}_default --model $m
  bench ${m}_wide2 --model $m --opt wide_gemm=2
  bench ${m}_wide2_k300 --model $m --opt wide_gemm=2 --opt tiled_min_k=300
  bench ${m}_wide2_k500 --model $m --opt wide_gemm=2 --opt tiled_min_k=500
done
exit 0
