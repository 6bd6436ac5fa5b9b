#!/bin/bash
# KS = 12 chain with the plain ring protocol (two iterations of DMA cover, no late waves) vs the wait-ahead protocol
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c29.log; : > $L
for n in 0 1 0 1; do
  lib=""; [ $n -gt 0 ] && lib=$repo/efficientconformer_amd/build/ab/libeffconf_ab$n.so
  echo "== variant $n" >> $L
  EFFCONF_ABLATE_LIB=$lib timeout 300 python tools/ablate_bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"median": [0-9.]*\|"ok": [a-z]*' | head -3 | tr '\n' ' ' >> $L
  echo >> $L
done
cat $L
