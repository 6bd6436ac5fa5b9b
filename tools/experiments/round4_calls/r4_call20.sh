#!/bin/bash
# round 4, call 20: the multi-rank code path on RCCL with ONE rank (bench.py --force-dist): what the per-range collective protocol costs on the
# real backend when nothing crosses xGMI - pipelined and sync - against the plain N = 1 step on the same box
set -u
out=gpurun_out/r4_20b; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for a in "" "--force-dist --pipeline-gather 1" "--force-dist --pipeline-gather 0" "--force-dist --pipeline-gather 1 --wire fp32"; do
  timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline $a 2>$out/err.txt | tee $out/raw.txt | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d.get('check', {})
print('%-50s %.2f M  %.3f ms  median %.3f  check %s %s' % ('$a', d['value'] / 1e6, d['ms_per_step'], d['config']['step_ms']['median'], c.get('ok'), c.get('rank0_rows_of_every_gathered_chunk_equal_their_rerun_alone_after_wire_rounding')))" | tee -a $out/force_dist.txt
  grep -v '^{' $out/raw.txt | head -3; tail -2 $out/err.txt | grep -v 'amdgpu\|hostname'
done
