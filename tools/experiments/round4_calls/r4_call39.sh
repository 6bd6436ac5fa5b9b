#!/bin/bash
# round 4, call 39: refill behind the next chunk's first GEMM (branch-free issue, peeled loop): tests, bench x 3, kernel durations, phases
# (PROF builds: phase "norms+misc" = first GEMM of the next chunk inside the loop, "ffn gemm1" = the refill's DMA issue, "ffn advance" = wait + barrier)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_39; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_encoder.py tests/test_gpu_round3.py -q -m gpu -x 2>&1 | tail -4 | tee $out/pytest.txt
tag=${1:-refill_mid}
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt; done
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1 > "$out/trace_$tag.log" 2>&1 )
db=$(find /tmp/kt_$tag -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" "$out/kernel_stats_$tag.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 4 --warmup 1" > /dev/null
grep "chain_kernel<" $out/kernel_stats_$tag.txt | cut -c1-70,110-200 | tee -a $out/ab.txt
for k in 162 163; do
  echo "== EFFCONF_CHAIN_PHASES=$k" | tee -a $out/chain_phases.txt
  EFFCONF_CHAIN_PHASES=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 2>&1 | grep "chain phases" | tee -a $out/chain_phases.txt
done
