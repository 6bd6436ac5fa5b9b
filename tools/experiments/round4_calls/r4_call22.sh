#!/bin/bash
# round 4, call 22: bench.py --force-dist now exports GPU_MAX_HW_QUEUES=8 itself; with eight queues, do more row ranges / streams pay at N = 1?
set -u
out=gpurun_out/r4_22; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  envs=$1; shift
  env $envs timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>$out/err.txt | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d.get('check', {})
print('%-28s %-50s %.2f M  %.3f ms  check %s' % ('$envs', '$*', d['value'] / 1e6, d['ms_per_step'], c.get('ok')))" | tee -a $out/variants.txt
}
run "X=0" --force-dist
run "X=0" --force-dist --pipeline-gather 0
run "GPU_MAX_HW_QUEUES=16" --force-dist
run "GPU_MAX_HW_QUEUES=8" --streams 4
run "GPU_MAX_HW_QUEUES=8" --streams 5
run "GPU_MAX_HW_QUEUES=8" --streams 6
run "GPU_MAX_HW_QUEUES=8" --streams 4 --batch 384
run "X=0" --batch 384
