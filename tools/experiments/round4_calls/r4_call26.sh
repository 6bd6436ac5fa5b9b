#!/bin/bash
# round 4, call 26: s_memtime phase profile of relpos_attention2_kernel (EFFCONF_ATTN2_PHASES=1) in the default step (three ranges in flight) and on one stream
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_26; mkdir -p $out
for a in "" "--streams 1 --ranges 1"; do
  echo "== EFFCONF_ATTN2_PHASES=1 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 $a" | tee -a $out/attn2_phases.txt
  EFFCONF_ATTN2_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 $a 2>&1 | grep "attn2 phases\|^{" | cut -c1-300 | tee -a $out/attn2_phases.txt
done
