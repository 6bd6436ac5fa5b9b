#!/bin/bash
# round 5, call 17: timing-only ablations of relpos_attention2_kernel (EFFCONF_ATTN_ABLATE: 1 no compute, 2 no loads after the first block, 4 no stores, 8 no utterance search)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_17b; mkdir -p $out
trace() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 --streams 1 --ranges 3 > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "x" > /dev/null
  echo "ablate $tag" | tee -a $out/ab.txt
  grep "attention" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
for a in 15 31 63 127 255; do EFFCONF_ATTN_ABLATE=$a trace $a; done
exit 0
