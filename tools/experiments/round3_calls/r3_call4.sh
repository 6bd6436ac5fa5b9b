#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c4"; mkdir -p "$out"
timeout 1800 python -m pytest tests -q -m gpu > "$out/t_all.log" 2>&1; echo "all gpu tests rc=$?" | tee -a "$out/summary.txt"
tail -8 "$out/t_all.log"
