#!/bin/bash
# ablations of the PIPELINED KS = 16 FFN loop: average dispatch durations of head (kind 2) / tail (kind 3), default 3 ranges
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c21.log; : > $L
cd /tmp && export TMPDIR=/tmp
for n in 0 1 2 3 5 6 7; do
  lib=""; [ $n -gt 0 ] && lib=$repo/efficientconformer_amd/build/ab/libeffconf_ab$n.so
  rm -rf /tmp/ab$n
  EFFCONF_ABLATE_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ab$n -o run -- python $repo/tools/ablate_bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 > /tmp/ab$n.log 2>&1
  db=$(find /tmp/ab$n -name "*.db" | head -1)
  for k in "chain_kernel<16, 4, 3, 2" "chain_kernel<16, 4, 3, 3"; do
    python $repo/tools/rocprof_dispatches.py "$db" "$k" | awk -v n=$n -v k="$k" '{s+=$1; c++} END{printf "ablation %d  %s  calls %d  avg %.1f us\n", n, k, c, s/c}' >> $L
  done
done
cat $L
