#!/bin/bash
# round 5, call 2: s_memtime phase profiles of the column-pair chains (EFFCONF_CHAIN2_PHASES)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_02; mkdir -p $out
for v in "161 256" "162 192" "163 192" "121 192" "160 192"; do set -- $v
  echo "== EFFCONF_CHAIN2_PHASES=$1 chain_full_max=$2" | tee -a $out/phases.txt
  EFFCONF_CHAIN2_PHASES=$1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_full_max=$2 --opt chain_pair=2 2>&1 | grep "chain2 phases" | tee -a $out/phases.txt
done
echo "== one stream, one range" | tee -a $out/phases.txt
EFFCONF_CHAIN2_PHASES=161 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_full_max=256 --opt chain_pair=2 --streams 1 --ranges 1 2>&1 | grep "chain2 phases" | tee -a $out/phases.txt
exit 0
