#!/bin/bash
mkdir -p gpurun_out/r3_04; L=gpurun_out/r3_04/pytest_gpu.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $L
timeout 600 python bench.py --model ConformerCTCLarge --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_04/ConformerCTCLarge_bench.json 2>/dev/null
timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default bench', d['value'], d['ms_per_step'], d['check']['ok'])" >> $L
cat $L
