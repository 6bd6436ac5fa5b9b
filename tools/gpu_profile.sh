#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> [bench.py args...]
# rocprofv3 --kernel-trace --stats of bench.py -> gpurun_out/<tag>/kernel_stats.txt (copy the ones to keep into profiles/).
# Profiled with --no-check: the self-check's single-utterance forwards would dilute every per-kernel average (round 3's summaries were).
set -u
tag=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out/$tag"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check "$@" > "$repo/gpurun_out/$tag/prof.log" 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python "$repo/tools/rocprof_summary.py" "$db" "$repo/gpurun_out/$tag/kernel_stats.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check $*" > /dev/null
