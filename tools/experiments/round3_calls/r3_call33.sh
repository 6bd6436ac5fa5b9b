#!/bin/bash
mkdir -p gpurun_out; timeout 600 python tools/graph_step_probe.py 40 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/c33.log; cat gpurun_out/c33.log
