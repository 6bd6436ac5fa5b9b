#!/bin/bash
# round 4, call 4: which kernels run in the split mode, and how long (rocprofv3 --kernel-trace --stats)
set -u
tools/gpu_profile.sh r4_04_split --precision split --steps 3 --warmup 1 --no-check
head -40 gpurun_out/r4_04_split/kernel_stats.txt
tail -5 gpurun_out/r4_04_split/prof.log
