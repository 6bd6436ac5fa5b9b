#!/bin/bash
# round 4, call 19: the multi-rank worker with the pipelined leg (two ranks over gloo) and on RCCL with one rank
set -u
timeout 1200 python -m pytest tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -15
