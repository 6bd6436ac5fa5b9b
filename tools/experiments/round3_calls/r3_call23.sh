#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c23.log; : > $L
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "tiled_attention" 2>&1 | tail -5 >> $L
echo "== bench fp32" >> $L
timeout 600 python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/c23_fp32.json 2>> $L
python - >> $L <<'PY'
import json
d = json.load(open('gpurun_out/c23_fp32.json'))
print(d['value'], d['ms_per_step'], d['check']['ok'], d['check'].get('label_sequences_identical_to_oracle'))
for c in d['roofline'].get('kernel_classes', []):
    print(c)
PY
cat $L
