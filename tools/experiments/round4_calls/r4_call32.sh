#!/bin/bash
# round 4, call 32: workgroup dispatch rate (tools/dispatch_rate_probe.py)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_32; mkdir -p $out
timeout 300 python tools/dispatch_rate_probe.py 2>&1 | grep -v "amdgpu.ids" | tee $out/dispatch_rate.txt
