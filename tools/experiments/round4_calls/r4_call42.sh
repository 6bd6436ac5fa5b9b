#!/bin/bash
# round 4, call 42: lengths_ragged_kernel on 16 waves - ragged tests, bench, kernel duration
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_42; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x 2>&1 | tail -3 | tee $out/pytest.txt
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('lengths16', d['value'], d['ms_per_step'], d.get('check',{}).get('ok'))" | tee -a $out/ab.txt; done
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_f && rocprofv3 --kernel-trace --stats -d /tmp/kt_f -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 5 --warmup 2 > "$out/trace.log" 2>&1 )
db=$(find /tmp/kt_f -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" "$out/kernel_stats.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 5 --warmup 2" > /dev/null
grep "lengths_ragged\|chain_kernel<16\|sum of" $out/kernel_stats.txt | cut -c1-70,110-200
