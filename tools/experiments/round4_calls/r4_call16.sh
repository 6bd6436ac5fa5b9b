#!/bin/bash
# round 4, call 16: N tiles fastest in the grid (resident workgroups share their A rows)
set -u
out=gpurun_out/r4_16; mkdir -p $out
timeout 300 python tools/sx_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/sx_gemm_bench.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "split or exact or precision" 2>&1 | tail -2
timeout 600 python bench.py --precision split --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('split', d['value'], d['ms_per_step'], d['check']['ok'], d['check']['max_abs_err_vs_oracle'])"
