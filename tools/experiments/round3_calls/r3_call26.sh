#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c26.log; : > $L
timeout 900 python bench.py --model EfficientConformerTransducerMedium --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c26_transducer.json 2>> $L
python - >> $L <<'PY'
import json
d = json.load(open('gpurun_out/c26_transducer.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('check')))
PY
cat $L
