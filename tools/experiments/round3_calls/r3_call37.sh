#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c37.log; : > $L
timeout 900 python -m pytest tests -m gpu -x -q -s -k "ctc_head or ctc or fp32_mode_keeps" 2>&1 | grep -v amdgpu | tail -16 >> $L
for o in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --opt ctc_mfma=$o 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['kernel_classes']; print('ctc_mfma=$o', round(d['value']/1e6,2), round(d['ms_per_step'],3), d['check']['ok'], d['check']['label_sequences_identical_to_oracle'], d['check']['argmax_flips_vs_oracle'], {k: round(v['ms_per_step'],3) for k,v in c.items() if k in ('misc','ctc','head')})" >> $L
done
cat $L
