#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c24.log; : > $L
for args in "0.0 256" "1.2 96" "3.0 40"; do timeout 300 python tools/rnnt_diag.py $args 2>&1 | grep -v amdgpu.ids >> $L; done
timeout 900 python -m pytest tests -m gpu -x -q -k "rnnt or transducer or Transducer" 2>&1 | tail -6 >> $L
cat $L
