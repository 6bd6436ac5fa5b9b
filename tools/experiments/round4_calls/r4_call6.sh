#!/bin/bash
# round 4, call 6: sx_gemm_kernel after the pipeline / epilogue rewrite, scores kernel walking the key tiles; parity tests of the modes; profile
set -u
out=gpurun_out/r4_06; mkdir -p $out
timeout 300 python tools/sx_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/sx_gemm_bench.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "exact or split or precision or attention_maps or empty_row" 2>&1 | tail -5
tools/gpu_profile.sh r4_06_split --precision split --steps 3 --warmup 1 --no-check
head -24 gpurun_out/r4_06_split/kernel_stats.txt | cut -c1-60,110-170
timeout 600 python bench.py --precision split --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('split', d['value'], d['ms_per_step'], d['check'])"
