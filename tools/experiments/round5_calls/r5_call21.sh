#!/bin/bash
# round 5, call 21: full GPU suite after the hygiene changes + default bench
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_21; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $out/pytest.txt
timeout 300 python bench.py 2>$out/bench.err | grep '^{' > $out/bench.json; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['check'].get('ok'), d['check'].get('label_exact_requirement'), d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['kernel_classes'].items(): print('  ', k, round(v['ms_per_step'],3), v['bound'], round(v['frac'],3))"
exit 0
