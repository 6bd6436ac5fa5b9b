#!/bin/bash
# round 5, call 13: chain3 / chain2 alone on the chip (one stream) - phase profiles and durations without neighbour kernels
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_13; mkdir -p $out
trace() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 "$@" > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $*" > /dev/null
  grep "chain3\|chain2_kernel<16" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
trace p5_s1 --opt chain_pair=5 --streams 1 --ranges 3
trace p4_s1 --opt chain_pair=4 --streams 1 --ranges 3
trace p5_s1_pf6 --opt chain_pair=5 --opt chain_nt=96 --streams 1 --ranges 3
echo "== EFFCONF_CHAIN3_PHASES=3 one stream" | tee -a $out/phases.txt
EFFCONF_CHAIN3_PHASES=3 timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_pair=5 --streams 1 --ranges 3 2>&1 | grep "chain3 phases" | tee -a $out/phases.txt
echo "== EFFCONF_CHAIN3_PHASES=3 one stream, prefetch 6" | tee -a $out/phases.txt
EFFCONF_CHAIN3_PHASES=3 timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 6 --warmup 2 --opt chain_pair=5 --opt chain_nt=96 --streams 1 --ranges 3 2>&1 | grep "chain3 phases" | tee -a $out/phases.txt
exit 0
