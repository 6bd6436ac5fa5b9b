#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c39.log; : > $L
timeout 600 python tools/robustness_sweep.py 2>&1 | grep -v amdgpu.ids | tail -12 >> $L
timeout 600 python tools/poison_sweep.py 2>&1 | grep -v amdgpu.ids | tail -12 >> $L
cat $L
