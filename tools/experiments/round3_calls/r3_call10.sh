#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c10"; mkdir -p "$out"
timeout 300 python tools/overlap_probe.py > "$out/overlap_events.txt" 2>&1; grep -v amdgpu.ids "$out/overlap_events.txt"
timeout 300 python tools/overlap_probe.py --no-collective > "$out/overlap_events_nocoll.txt" 2>&1; grep -v amdgpu.ids "$out/overlap_events_nocoll.txt" | grep "rows\|next"
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -3
