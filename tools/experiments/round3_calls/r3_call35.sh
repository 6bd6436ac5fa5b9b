#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/c35.log; : > $L
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $L
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['check']['ok'])" >> $L
cat $L
