#!/bin/bash
# round 4, call 7: flat conv2d kernel, PV stats with one exponential per score; parity tests of the modes; profile
set -u
timeout 900 python -m pytest tests -m gpu -q -x -k "exact or split or precision or attention_maps or empty_row or every_shipped" 2>&1 | tail -5
tools/gpu_profile.sh r4_07_split --precision split --steps 3 --warmup 1 --no-check
head -14 gpurun_out/r4_07_split/kernel_stats.txt | cut -c1-60,110-170
timeout 600 python bench.py --precision split --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('split', d['value'], d['ms_per_step'], d['check']['ok'], d['check']['max_abs_err_vs_oracle'])"
timeout 600 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('fp32', d['value'], d['ms_per_step'], d['check']['ok'], d['check']['max_abs_err_vs_oracle'])"
