#!/bin/bash
# round 5, call 18: the full GPU suite on chain_pair = 5; the --ragged 0 roofline leg (gemm_other 14.6 ms in a 5.9 ms step)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_18; mkdir -p $out
timeout 300 python bench.py --no-cpu-baseline --no-check --ragged 0 --steps 5 --warmup 2 2>&1 | grep '^{' > $out/bench_ragged0.json
python - <<PY
import json
d=json.load(open('$out/bench_ragged0.json'))
print('ragged0 ms_per_step', d['ms_per_step'])
for k,v in d['roofline']['kernel_classes'].items(): print('  ', k, round(v['ms_per_step'],3), v['launches_per_step'])
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_r0 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_r0 -o run -- python "$repo/bench.py" --no-cpu-baseline --no-check --ragged 0 --steps 2 --warmup 1 > "$out/trace_r0.log" 2>&1 )
db=$(find /tmp/kt_r0 -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" "$out/kernel_stats_ragged0_with_roofline_leg.txt" "python bench.py --no-cpu-baseline --no-check --ragged 0 --steps 2 --warmup 1" > /dev/null
head -24 $out/kernel_stats_ragged0_with_roofline_leg.txt | cut -c1-90,110-200
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $out/pytest.txt
exit 0
