#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c24.log; : > $L
for args in "0.0 256" "1.2 96"; do RNNT_MODES=0:0,1:1 timeout 300 python tools/rnnt_diag.py $args 2>&1 | grep -v amdgpu.ids >> $L; done
cat $L
