#!/bin/bash
# round 4, call 43: robustness sweeps on the final tree (row-range splits cross the chain_small_m threshold: 2-wave and 8-wave chain shapes in one comparison)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_43; mkdir -p $out
timeout 600 python tools/robustness_sweep.py 2>&1 | grep -v amdgpu.ids | tee $out/robustness_sweep.txt | tail -6
timeout 600 python tools/poison_sweep.py --reps 10 2>&1 | grep -v amdgpu.ids | tee $out/poison_sweep.txt | tail -8
