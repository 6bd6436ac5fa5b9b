#!/bin/bash
# round 4, call 30: attention with three workgroups per CU at head widths <= 64 (8 KiB swizzled V tile, one staging set) against two (--opt attn_waves=2)
set -u
repo=$(pwd); out=$repo/gpurun_out/r4_30; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_encoder.py -q -m gpu -x -k "ragged or attention or stream or causal or band" 2>&1 | tail -4 | tee $out/pytest.txt
for i in 1 2 3; do for o in "" "--opt attn_waves=2"; do timeout 300 python bench.py --no-cpu-baseline --no-roofline $o 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$o', d['value'], d['ms_per_step'], d.get('check',{}).get('ok'))" | tee -a $out/att_3wg.txt; done; done
EFFCONF_ATTN2_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-check --steps 10 --warmup 3 2>&1 | grep "attn2 phases" | cut -c1-200 | tee $out/attn2_phases.txt
