#!/bin/bash
# round 5, call 26: the one-rank RCCL path (--force-dist: process group, per-range all-gather, pipelined head) against the plain step, chain_pair 4 / 5
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_26; mkdir -p $out
bench() {
  tag=$1; shift
  for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench plain
bench dist --force-dist
bench dist_p4 --force-dist --opt chain_pair=4
bench dist_p0 --force-dist --opt chain_pair=0 --opt chain_variant=0
bench dist_sync --force-dist --pipeline-gather 0
bench dist_labels --force-dist --gather labels
exit 0
