#!/bin/bash
# round 5, call 30: attention2 prologue - first key block's K / V loads and the positional band's loads issued before the query rows are converted, the
# ablation mask compiled out of the product kernel: tests that touch the kernel, then the step
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_30; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "attention or ragged or stream or golden" < /dev/null 2>&1 | tail -5 | tee $out/pytest.txt
bench() {
  tag=$1; shift
  for i in 1 2 3 4; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" < /dev/null 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench default
exit 0
