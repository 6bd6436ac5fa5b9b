#!/bin/bash
# round 5, call 8: the full GPU suite on the new defaults (column-pair chains at padded width 256, 4-wave D <= 128 chains, chunk-major FFN weights) + bench
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_08; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $out/pytest.txt
timeout 300 python bench.py 2>&1 | grep '^{' > $out/bench.json; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['check'].get('ok'), d['roofline']['frac'])"
exit 0
