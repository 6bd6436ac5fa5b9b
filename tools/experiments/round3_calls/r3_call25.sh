#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c25.log; : > $L
timeout 900 python -m pytest tests -m gpu -x -q -k "rnnt or transducer or Transducer" 2>&1 | tail -4 >> $L
timeout 600 python bench.py --model EfficientConformerTransducerMedium --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c25_transducer.json 2>> $L
python - >> $L <<'PY'
import json
d = json.load(open('gpurun_out/c25_transducer.json'))
print(d['value'], d['ms_per_step'], d.get('check', {}).get('ok'))
print(json.dumps(d.get('transducer_legs')))
PY
timeout 300 python tools/rnnt_diag.py 0.0 256 2>&1 | grep -v amdgpu.ids >> $L
cat $L
