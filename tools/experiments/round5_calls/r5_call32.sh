#!/bin/bash
# round 5, call 32: the parts EVIDENCE_TRIM left out of r5_31, on the same tree: SQ counters, the round-1 / B = 128 workloads, the all-gather stand-in
set -u
tag=r5_31
repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_f && timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_f -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 < /dev/null > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_f -name "*.db" | head -1)
[ -n "$db" ] && python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1" > /dev/null
timeout 200 python bench.py --no-trim --no-cpu-baseline --no-roofline < /dev/null > "$out/bench_notrim.json" 2>> "$out/bench.err"
timeout 200 python bench.py --batch 128 --no-cpu-baseline --no-roofline < /dev/null > "$out/bench_b128.json" 2>> "$out/bench.err"
for args in "--mode sync" "--mode pipelined" "--no-collective"; do
  echo "## overlap_probe.py --wire bf16 $args" >> "$out/overlap_events.txt"
  timeout 200 python tools/overlap_probe.py --wire bf16 $args < /dev/null 2>/dev/null | grep -v amdgpu.ids >> "$out/overlap_events.txt"
done
timeout 200 python bench.py --force-dist --no-cpu-baseline --no-roofline < /dev/null > "$out/bench_one_rank_rccl.json" 2>> "$out/bench.err"
ls -la $out | tail -8
exit 0
