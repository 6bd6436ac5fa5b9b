#!/bin/bash
# absolute per-dispatch durations of the KS = 16 chains, 1 and 3 streams
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c19.log; : > $L
cd /tmp && export TMPDIR=/tmp
for st in 1 3; do
  rm -rf /tmp/q$st
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/q$st -o run -- python $repo/bench.py --no-cpu-baseline --no-roofline --no-check --steps 3 --warmup 1 --streams $st > /tmp/q$st.log 2>&1
  db=$(find /tmp/q$st -name "*.db" | head -1)
  echo "== streams $st" >> $L
  grep -o '"ms_per_step": [0-9.]*' /tmp/q$st.log >> $L
  for k in "chain_kernel<16, 4, 3, 2" "chain_kernel<16, 4, 3, 3" "chain_kernel<16, 4, 3, 0" "chain_kernel<12, 8, 3, 1" "chain_kernel<8, 8, 4, 1"; do
    python $repo/tools/rocprof_dispatches.py "$db" "$k" | tail -18 | cut -c1-60 >> $L
  done
done
cat $L
