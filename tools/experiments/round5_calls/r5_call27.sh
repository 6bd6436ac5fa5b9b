#!/bin/bash
# round 5, call 27: attention2 with the skew applied by the writer (64-float rows: 52 KB of LDS at head width 64) - default kernels and the
# three-workgroups-per-CU variant (attn_waves = 3: one staging set, 168 registers)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_27; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or ragged or stream or golden" 2>&1 | tail -5 | tee $out/pytest.txt
bench() {
  tag=$1; shift
  for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench default
bench occ3 --opt attn_waves=3
bench default
bench occ3 --opt attn_waves=3
cd /tmp && export TMPDIR=/tmp
for v in 4 3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$v -o t -- python $repo/bench.py --no-cpu-baseline --no-roofline --no-check --steps 5 --warmup 2 --streams 1 --ranges 1 --opt attn_waves=$v > $out/prof_$v.log 2>&1
  f=$(ls $out/prof_$v/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $out/prof_$v/*kernel_stats.csv | head -1)
  echo "== attn_waves=$v (one stream, one range)" >> $out/kernels.txt; grep -i "attention\|Name" $f | cut -c1-200 >> $out/kernels.txt
done
cat $out/kernels.txt
exit 0
