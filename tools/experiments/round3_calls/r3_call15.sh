#!/bin/bash
# ablations of the KS = 16 FFN loop: per-kernel average durations (rocprofv3 --kernel-trace --stats), one stream
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c15.log; : > $L
cd /tmp && export TMPDIR=/tmp
for n in 0 1 2 3 4 5; do
  lib=""; [ $n -gt 0 ] && lib=$repo/efficientconformer_amd/build/ab/libeffconf_ab$n.so
  rm -rf /tmp/ab$n
  EFFCONF_ABLATE_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ab$n -o run -- python $repo/tools/ablate_bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --streams 1 > /tmp/ab$n.log 2>&1
  db=$(find /tmp/ab$n -name "*.db" | head -1)
  echo "== ablation $n" >> $L
  grep -o '"ms_per_step": [0-9.]*' /tmp/ab$n.log >> $L
  python $repo/tools/rocprof_summary.py "$db" /tmp/ab$n.txt x | grep "chain_kernel<16" | sed 's/void (anonymous namespace):://; s/(anonymous namespace):://g' | cut -c1-40,100-160 >> $L
done
cat $L
