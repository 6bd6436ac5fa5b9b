#!/bin/bash
# round 4, call 33: small-batch latency by kernel route (chains against per-GEMM kernels) + kernel durations at B = 4
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_33; mkdir -p $out
timeout 600 python tools/small_batch_routes.py 2>&1 | grep -v "amdgpu.ids" | tee $out/small_batch_routes.txt
for v in a b; do
  o=""; [ $v = b ] && o="--opts fuse_chain=0"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sb_$v && rocprofv3 --kernel-trace --stats -d /tmp/sb_$v -o run -- python "$repo/tools/small_batch_routes.py" --only 4 $o > "$out/trace_$v.log" 2>&1 )
  db=$(find /tmp/sb_$v -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_b4_$v.txt" "python tools/small_batch_routes.py --only 4 $o --steps 20 --warmup 5" > /dev/null
  echo "== $v $o"; head -24 $out/kernel_stats_b4_$v.txt | cut -c1-70,110-190
done
