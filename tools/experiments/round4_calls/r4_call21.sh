#!/bin/bash
# round 4, call 21: why does the multi-rank path cost +32 % on RCCL with one rank?  stream counts / hardware queues / gather modes
set -u
out=gpurun_out/r4_21; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # <env assignments> -- <bench args>
  envs=$1; shift
  env $envs timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>$out/err.txt | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d.get('check', {})
print('%-28s %-62s %.2f M  %.3f ms  check %s' % ('$envs', '$*', d['value'] / 1e6, d['ms_per_step'], c.get('ok')))" | tee -a $out/force_dist_variants.txt
}
run "X=0" --force-dist
run "X=0" --force-dist --streams 2
run "X=0" --force-dist --streams 2 --ranges 3
run "X=0" --force-dist --gather labels
run "GPU_MAX_HW_QUEUES=8" --force-dist
run "GPU_MAX_HW_QUEUES=8"
run "X=0" --streams 2
run "TORCH_NCCL_HIGH_PRIORITY=1" --force-dist
