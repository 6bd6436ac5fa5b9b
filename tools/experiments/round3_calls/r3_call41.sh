#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c41.log; : > $L
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "two_layer" 2>&1 | tail -6 >> $L
for r in 0 1; do
  timeout 600 python bench.py --model ConformerCTCLarge --steps 5 --warmup 2 --no-cpu-baseline --ragged $r 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('ConformerCTCLarge ragged=$r', round(d['value']/1e6,3), round(d['ms_per_step'],2), d['check']['ok'], d['check'].get('argmax_flips_vs_oracle'))" >> $L
done
cat $L
