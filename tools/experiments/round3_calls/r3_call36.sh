#!/bin/bash
# bench.py's own multi-rank path on one GPU: two ranks over gloo, both on cuda:0 (never a benchmark - a launch-path check)
mkdir -p gpurun_out; L=gpurun_out/c36.log; : > $L
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --one-device --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -6 | cut -c1-900 >> $L
echo "== self-spawn (python bench.py --gpus 2)" >> $L
timeout 600 python bench.py --gpus 2 --backend gloo --one-device --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -4 | cut -c1-900 >> $L
cat $L
