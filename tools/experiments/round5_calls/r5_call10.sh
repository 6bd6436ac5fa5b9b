#!/bin/bash
# round 5, call 10: phase profile of the role-split chain (chain3.hip)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_10; mkdir -p $out
for k in 3 1; do
  echo "== EFFCONF_CHAIN3_PHASES=$k" | tee -a $out/phases.txt
  EFFCONF_CHAIN3_PHASES=$k timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_pair=5 2>&1 | grep "chain3 phases" | tee -a $out/phases.txt
done
exit 0
