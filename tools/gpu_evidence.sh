#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_evidence.sh <tag>
# One box, one call: everything profiles/<tag>_* is made of.  Outputs under gpurun_out/<tag>/ (copy the ones to keep into profiles/).
#   pytest_gpu.txt                   the whole -m gpu suite + smoke()
#   bench.json                       python bench.py (the driver's command: roofline on the timed launch shape + full-batch shape, cpu_baseline, check)
#   kernel_stats.txt                 rocprofv3 --kernel-trace --stats of the same command, --no-check --no-roofline (only the timed region's launches)
#   pmc_hbm_traffic.txt, pmc_traffic.json   two --pmc passes (FETCH_SIZE, WRITE_SIZE), clean as above; second table = serialised kernel durations
#   sq_counters.txt                  one --pmc pass of SQ counters (MFMA busy / VALU / LDS wait per kernel)
#   bench_split.json, bench_fp32.json  the two label-exact modes; <model>_bench_split.json, split_kernel_stats.txt, split_sq_counters.txt (round 6)
#   <model>_bench.json               bench.py --model ... for the other configurations of BASELINE.json
#   bench_ragged0.json, bench_notrim.json, bench_b128.json   the round-2 / round-1 workloads and B = 128 on the final tree
#   overlap_events.txt               tools/overlap_probe.py (one-GPU stand-in of the per-range all-gather): sync / pipelined / no collective
#   large_kernel_stats.txt           kernel trace of EfficientConformerCTCLarge
set -u
tag=$1
repo=$(pwd)
out="$repo/gpurun_out/$tag"
mkdir -p "$out"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > "$out/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> "$out/pytest_gpu.txt"
python bench.py --steps 50 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"
tools/gpu_profile.sh "$tag" --steps 5 --warmup 2
tools/gpu_pmc.sh "$tag" --steps 5 --warmup 2
if [ -z "${EVIDENCE_TRIM:-}" ]; then   # EVIDENCE_TRIM=1: without the SQ-counter pass, the round-1 / B = 128 workloads and the overlap probe (GPU-minutes)
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_f && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_f -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_f -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1" > /dev/null
fi
python bench.py --precision split --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_split.json" 2>> "$out/bench.err"
python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_fp32.json" 2>> "$out/bench.err"
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge; do      # round 6: the label-exact mode on the other CTC configurations (labels vs the oracle in `check`)
  python bench.py --precision split --model $m --steps 5 --warmup 2 --no-cpu-baseline > "$out/${m}_bench_split.json" 2>> "$out/bench.err"
done
tools/prof_split.sh "${tag}_split" pmc
cp "$repo/gpurun_out/${tag}_split/kernel_stats.txt" "$out/split_kernel_stats.txt"; cp "$repo/gpurun_out/${tag}_split/sq_counters.txt" "$out/split_sq_counters.txt" 2>/dev/null
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge ConformerCTCLarge EfficientConformerTransducerMedium; do
  python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > "$out/${m}_bench.json" 2> "$out/${m}_bench.err"
done
python bench.py --ragged 0 --no-cpu-baseline > "$out/bench_ragged0.json" 2>> "$out/bench.err"
if [ -z "${EVIDENCE_TRIM:-}" ]; then
python bench.py --no-trim --no-cpu-baseline --no-roofline > "$out/bench_notrim.json" 2>> "$out/bench.err"
python bench.py --batch 128 --no-cpu-baseline --no-roofline > "$out/bench_b128.json" 2>> "$out/bench.err"
for args in "--mode sync" "--mode pipelined" "--no-collective"; do
  echo "## overlap_probe.py --wire bf16 $args" >> "$out/overlap_events.txt"
  python tools/overlap_probe.py --wire bf16 $args 2>/dev/null | grep -v amdgpu.ids >> "$out/overlap_events.txt"
done
fi
tools/gpu_profile.sh "${tag}_large" --model EfficientConformerCTCLarge --steps 3 --warmup 1
cp "$repo/gpurun_out/${tag}_large/kernel_stats.txt" "$out/large_kernel_stats.txt"
ls -la "$out"; cat "$out/pytest_gpu.txt"
