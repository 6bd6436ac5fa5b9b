#!/bin/bash
# round 5, call 28: three attention workgroups per CU at head width 64 as the default + the wave-parallel utterance search (attention, depthwise conv):
# the tests that touch them, then the step against attn_waves = 2 (two staging sets, two workgroups per CU)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_28; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "attention or ragged or dw or conv or stream or golden or round5" < /dev/null 2>&1 | tail -5 | tee $out/pytest.txt
bench() {
  tag=$1; shift
  for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" < /dev/null 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench default
bench two_sets --opt attn_waves=2
bench default
exit 0
