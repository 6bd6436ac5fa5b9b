#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_evidence.sh <tag>
# One box, one call: everything profiles/<tag>_* is made of.  Outputs under gpurun_out/<tag>/ (copy the ones to keep into profiles/).
#   bench.json                       python bench.py (the driver's command: roofline + cpu_baseline + check)
#   kernel_stats.txt                 rocprofv3 --kernel-trace --stats of the same command
#   pmc_hbm_traffic.txt, pmc_traffic.json   two --pmc passes (FETCH_SIZE, WRITE_SIZE)
#   sq_counters.txt                  one --pmc pass of SQ counters (MFMA busy / VALU / LDS wait per kernel)
#   <model>_bench.json               bench.py --model ... for the other configurations of BASELINE.json
#   bench_ragged0.json, bench_notrim.json, bench_b128.json, bench_fp32.json   the round-2 workloads and the label-exact mode of the final tree
#   overlap_events.txt               tools/overlap_probe.py (one-GPU stand-in of the per-range all-gather)
#   small_batch_latency.txt          B = 1 / 4 / 16 eager vs hipGraph replay
#   large_kernel_stats.txt, large_sq_counters.txt   the same two profiles for EfficientConformerCTCLarge
set -u
tag=$1
repo=$(pwd)
out="$repo/gpurun_out/$tag"
mkdir -p "$out"
python bench.py --steps 50 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"
tools/gpu_profile.sh "$tag" --steps 5 --warmup 2
tools/gpu_pmc.sh "$tag" --steps 5 --warmup 2
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
sq() {   # <name> <bench args...>
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_$name && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_$name -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 "$@" > "$out/sq_$name.log" 2>&1 )
  db=$(find /tmp/sq_$name -name "*.db" | head -1)
  python tools/sq_summary.py "$db" "$out/${name}sq_counters.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 $*" > /dev/null
}
sq ""
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge ConformerCTCLarge EfficientConformerTransducerMedium; do
  python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > "$out/${m}_bench.json" 2> "$out/${m}_bench.err"
done
python bench.py --ragged 0 --no-cpu-baseline > "$out/bench_ragged0.json" 2>> "$out/bench.err"
python bench.py --no-trim --no-cpu-baseline --no-roofline > "$out/bench_notrim.json" 2>> "$out/bench.err"
python bench.py --batch 128 --no-cpu-baseline --no-roofline > "$out/bench_b128.json" 2>> "$out/bench.err"
python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 > "$out/bench_fp32.json" 2>> "$out/bench.err"
python tools/overlap_probe.py 2>/dev/null | grep -v amdgpu.ids > "$out/overlap_events.txt"
python tools/overlap_probe.py --no-collective 2>/dev/null | grep -v amdgpu.ids > "$out/overlap_events_nocollective.txt"
python tools/small_batch_latency.py 2>/dev/null | grep "B=" > "$out/small_batch_latency.txt"
tools/gpu_profile.sh "${tag}_large" --model EfficientConformerCTCLarge --steps 3 --warmup 1
cp "$repo/gpurun_out/${tag}_large/kernel_stats.txt" "$out/large_kernel_stats.txt"
sq large_ --model EfficientConformerCTCLarge
ls -la "$out"
