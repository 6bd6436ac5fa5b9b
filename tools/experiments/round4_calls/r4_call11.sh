#!/bin/bash
# round 4, call 11: kernel trace of EfficientConformerCTCLarge (clean: --no-check) with the front end on ragged rows
set -u
tools/gpu_profile.sh r4_11_large --model EfficientConformerCTCLarge --steps 3 --warmup 1
head -34 gpurun_out/r4_11_large/kernel_stats.txt | cut -c1-64,110-190
