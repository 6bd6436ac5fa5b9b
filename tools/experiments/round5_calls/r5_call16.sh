#!/bin/bash
# round 5, call 16: attention2 - idle waves of an utterance's last query tile skip the compute, 32-key blocks for short tails: parity tests, bench, kernel durations
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_16; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_exact_and_sweep.py tests/test_gpu_round3.py -q -m gpu -x -k "attention or ragged or streaming or maps" 2>&1 | tail -5 | tee $out/pytest.txt
bench() {
  tag=$1; shift
  for i in 1 2 3; do timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
trace() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 "$@" > "$out/trace_$tag.log" 2>&1 )
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 $*" > /dev/null
  grep "attention" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
}
bench p4
bench p5 --opt chain_pair=5
trace p4
exit 0
