#!/bin/bash
# round 4, call 31: kernel durations (rocprofv3 --kernel-trace --stats) of the attention variants: three workgroups per CU (default now) against two
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_31; mkdir -p $out
for v in a b; do
  o=""; [ $v = b ] && o="--opt attn_waves=2"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$v && rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $o > "$out/trace_$v.log" 2>&1 )
  db=$(find /tmp/kt_$v -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$v.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $o" > /dev/null
  echo "== variant $v ($o)"; grep "attention2\|sum of kernel\|library kernels" $out/kernel_stats_$v.txt | cut -c1-60,110-200
done
