#!/bin/bash
# round 4, call 37: s_memtime phase profiles of the pipelined chains (EFFCONF_CHAIN_PHASES = 162: D = 240 tail, 163: D = 240 head, 81: D = 120 full chain)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_37; mkdir -p $out
for k in 162 163 81; do
  echo "== EFFCONF_CHAIN_PHASES=$k" | tee -a $out/chain_phases.txt
  EFFCONF_CHAIN_PHASES=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 2>&1 | grep "chain phases" | tee -a $out/chain_phases.txt
done
