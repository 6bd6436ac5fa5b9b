#!/bin/bash
# round 4, call 40: phase profile of the D = 240 head chain with the Q/K/V stage split (wait + barrier / refill / MFMAs / write-out)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_40; mkdir -p $out
for k in 162; do
  echo "== EFFCONF_CHAIN_PHASES=$k" | tee -a $out/chain_phases.txt
  EFFCONF_CHAIN_PHASES=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 2>&1 | grep "chain phases" | tee -a $out/chain_phases.txt
done
