#!/bin/bash
# round 4, call 23: NaN guards around every parameter buffer in the split mode too (all 14 configurations); poisoned workspaces / repeated forwards
set -u
timeout 1200 python -m pytest tests -m gpu -q -x -k "reads_past_a_parameter_buffer" 2>&1 | tail -4
timeout 600 python tools/poison_sweep.py 2>&1 | grep -v amdgpu | tail -12
