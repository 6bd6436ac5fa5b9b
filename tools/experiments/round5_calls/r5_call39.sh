#!/bin/bash
# round 5, call 39: the matrix-pipe depthwise kernel at kernel size 31 (option dwconv_mfma = 2, not the default) on ConformerCTC-Large and -Small
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_39; mkdir -p $out
for m in ConformerCTCLarge ConformerCTCSmall; do
for v in 1 2 1 2; do
  timeout 120 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-check --opt dwconv_mfma=$v < /dev/null 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$m dwconv_mfma=$v', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt
done; done
exit 0
