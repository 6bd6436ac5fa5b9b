#!/bin/bash
# round 4, call 29: step-boundary bubble - one against two / three batches in flight (tools/inflight_probe.py)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_29; mkdir -p $out
timeout 600 python tools/inflight_probe.py 2>&1 | grep -v "^\[" | tail -8 | tee $out/inflight.txt
