#!/bin/bash
mkdir -p gpurun_out; timeout 600 python tools/front_door_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c31.log; cat gpurun_out/c31.log
