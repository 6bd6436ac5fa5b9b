#!/bin/bash
# round 4, call 27: attention2 with the wave-parallel utterance lookup, all prologue loads in one round trip, branch-free block loads
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_27; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_encoder.py -q -m gpu -x -k "ragged or attention or stream or causal or band" 2>&1 | tail -4 | tee $out/pytest.txt
for a in "" "--streams 1 --ranges 1"; do
  echo "== EFFCONF_ATTN2_PHASES=1 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 $a" | tee -a $out/attn2_phases.txt
  EFFCONF_ATTN2_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 $a 2>&1 | grep "attn2 phases\|^{" | cut -c1-300 | tee -a $out/attn2_phases.txt
done
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('check',{}).get('ok'), {k:v for k,v in d.get('kernel_classes',{}).items()} if 0 else '')" | tee -a $out/lines.txt; done
