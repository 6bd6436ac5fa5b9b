#!/bin/bash
mkdir -p gpurun_out; ( timeout 600 python tools/front_door_bench.py 1024 256; timeout 900 python tools/front_door_bench.py 4096 256 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/c30.log; cat gpurun_out/c30.log
timeout 600 python -m pytest tests -m gpu -x -q -k "front_door" 2>&1 | tail -3
timeout 600 python tools/graph_step_probe.py 40 2>&1 | grep -v amdgpu.ids | tail -6
