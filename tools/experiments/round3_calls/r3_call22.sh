#!/bin/bash
# pipelined KS = 16 FFN loop, steady state peeled: default bench + check, dispatch durations, GPU tests
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c22.log; : > $L
echo "== default bench" >> $L
timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids | cut -c1-2400 >> $L
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p22 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p22 -o run -- python $repo/bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 > /tmp/p22.log 2>&1
  db=$(find /tmp/p22 -name "*.db" | head -1)
  for k in "chain_kernel<16, 4, 3, 2" "chain_kernel<16, 4, 3, 3" "chain_kernel<16, 4, 3, 0" "chain_kernel<12, 8, 3, 1"; do
    python $repo/tools/rocprof_dispatches.py "$db" "$k" | awk -v k="$k" '{s+=$1; c++} END{printf "%s  calls %d  avg %.1f us\n", k, c, s/c}' >> $L
  done )
echo "== pytest" >> $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 >> $L
cat $L
