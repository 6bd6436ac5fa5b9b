#!/bin/bash
# round 4, call 25: CTC head on bf16 rows (effconf_ctc_greedy_bf16) - bit identity test, the dist tests, and the one-rank RCCL line with it
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_25; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | tail -4 | tee $out/pytest.txt
for i in 1 2; do
  for a in "" "--force-dist"; do
    echo "== bench.py $a" | tee -a $out/lines.txt
    GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --no-cpu-baseline --no-roofline $a 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('check',{}).get('ok'))" | tee -a $out/lines.txt
  done
done
