#!/bin/bash
# pipelined KS = 16 FFN loop: default bench with its self-check, one-stream kernel stats, GPU tests
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c16.log; : > $L
echo "== default bench" >> $L
timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids | cut -c1-3000 >> $L
echo "== kernel stats, one stream" >> $L
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p16 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p16 -o run -- python $repo/bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --streams 1 > /tmp/p16.log 2>&1
  db=$(find /tmp/p16 -name "*.db" | head -1); python $repo/tools/rocprof_summary.py "$db" /tmp/p16.txt x | grep "chain_kernel" | sed 's/void (anonymous namespace):://; s/(anonymous namespace):://g' | cut -c1-36,100-160 >> $L )
echo "== pytest" >> $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $L
cat $L
