#!/bin/bash
# round 4, call 46: the multi-rank code path on RCCL with one rank (bench.py --force-dist) on the last tree: default, sync gather, fp32 wire, Large
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_46; mkdir -p $out
for a in "" "--force-dist" "--force-dist --pipeline-gather 0" "--force-dist --wire fp32" "--force-dist --model EfficientConformerCTCLarge --steps 5 --warmup 2"; do
  timeout 400 python bench.py --no-cpu-baseline --no-roofline $a 2>$out/err.txt | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d.get('check', {})
print('%-70s %.2f M  %.3f ms  check %s' % ('bench.py $a', d['value'] / 1e6, d['ms_per_step'], c.get('ok')))" | tee -a $out/rccl_one_rank.txt
done
