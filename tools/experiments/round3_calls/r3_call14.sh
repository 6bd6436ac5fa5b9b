#!/bin/bash
# KS = 16 batched fragment reads: phase profiles, default bench with its self-check, GPU tests
mkdir -p gpurun_out
L=gpurun_out/c14.log; : > $L
for k in 162 163; do
  echo "== EFFCONF_CHAIN_PHASES=$k" >> $L
  EFFCONF_CHAIN_PHASES=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $L
done
echo "== default bench" >> $L
timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids >> $L
echo "== pytest" >> $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 >> $L
cat $L
