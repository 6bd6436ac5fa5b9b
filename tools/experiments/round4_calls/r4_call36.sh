#!/bin/bash
# round 4, call 36: software-pipelined FFN stage in the D <= 128 chains (chain.hip, PIPE) - tests, then A / B on ONE box against the unpipelined file
# (kept for the run as tools/experiments/ab/chain_round3.hip.txt = git show f15a03e:efficientconformer_amd/csrc/chain.hip + the chain_small_m plumbing; removed afterwards), rebuilt in place on the box between the runs
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_36; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_encoder.py tests/test_gpu_round3.py -q -m gpu -x 2>&1 | tail -4 | tee $out/pytest.txt
run() {
  tag=$1
  for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt; done
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1" > /dev/null
  grep "chain_kernel<8\|chain_kernel<16" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
run pipelined
cp tools/experiments/ab/chain_round3.hip.txt efficientconformer_amd/csrc/chain.hip
python -m efficientconformer_amd._build 2>&1 | tail -1
run round3; exit 0
