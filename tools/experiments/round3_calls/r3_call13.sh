#!/bin/bash
# KS = 16 phase profiles (head 162, tail 163), default 3-range bench and one stream
mkdir -p gpurun_out
for k in 162 163; do
  for st in 3 1; do
    echo "== EFFCONF_CHAIN_PHASES=$k streams=$st" >> gpurun_out/c13.log
    EFFCONF_CHAIN_PHASES=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --streams $st >> gpurun_out/c13.log 2>&1
  done
done
tail -80 gpurun_out/c13.log
