#!/bin/bash
# round 5, call 14: timing-only ablations of the chain3 FFN loop (one stream): no first GEMM (256), no second GEMM (512), no weight stream (1024)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_14; mkdir -p $out
for nt in 0 256 512 768 1024 1280 1792; do
  echo "== chain_nt=$nt" | tee -a $out/phases.txt
  EFFCONF_CHAIN3_PHASES=3 timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_pair=5 --opt chain_nt=$nt --streams 1 --ranges 3 2>&1 | grep "chain3 phases" | grep -v SIMD | tee -a $out/phases.txt
done
exit 0
