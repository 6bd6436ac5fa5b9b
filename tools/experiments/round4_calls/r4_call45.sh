#!/bin/bash
# round 4, call 45: L2 -> LDS fill rate of a CU, LDS-DMA against loads + ds_write, by waves per workgroup and workgroups in flight (tools/lds_fill_rate_probe.py)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_45; mkdir -p $out
timeout 300 python tools/lds_fill_rate_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/lds_fill_rate.txt
