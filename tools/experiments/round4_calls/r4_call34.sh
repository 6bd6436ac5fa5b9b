#!/bin/bash
# round 4, call 34: 2-wave chain workgroups for small launches: bit identity, small-batch latency, the default line unchanged
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_34; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "two_wave" 2>&1 | tail -5 | tee $out/pytest.txt
timeout 600 python tools/small_batch_routes.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-330 | tee $out/small_batch_routes.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sb_c && rocprofv3 --kernel-trace --stats -d /tmp/sb_c -o run -- python "$repo/tools/small_batch_routes.py" --only 4 > "$out/trace_c.log" 2>&1 )
db=$(find /tmp/sb_c -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" "$out/kernel_stats_b4_small_m.txt" "python tools/small_batch_routes.py --only 4 --steps 20 --warmup 5" > /dev/null
head -24 $out/kernel_stats_b4_small_m.txt | cut -c1-70,110-190
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{' | cut -c1-260
