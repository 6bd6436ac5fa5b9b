#!/bin/bash
# round 4, call 44: is three ranges on three streams still the best arrangement on the final tree?
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_44; mkdir -p $out
for a in "--streams 3" "--streams 2" "--streams 4" "--streams 3 --ranges 6" "--streams 3 --range-frames 30,35,35" "--streams 3 --range-frames 36,32,32" "--streams 3 --batch 384" "--streams 3 --batch 512"; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check $a 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-44s %.2f M  %.3f ms' % ('$a', d['value']/1e6, d['ms_per_step']))" | tee -a $out/arrangements.txt
done
