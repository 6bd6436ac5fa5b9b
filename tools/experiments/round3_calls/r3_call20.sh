#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c20.log; : > $L
for k in 162 163; do
  echo "== EFFCONF_CHAIN_PHASES=$k" >> $L
  EFFCONF_CHAIN_PHASES=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $L
done
cat $L
