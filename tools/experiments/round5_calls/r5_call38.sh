#!/bin/bash
# round 5, call 38: the final tree (attention at three workgroups per CU + depthwise convolution on the matrix pipe): whole GPU suite, smoke, the driver's
# command, kernel trace, the other configurations
set -u
tag=r5_38
repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q < /dev/null 2>&1 | tail -4 > $out/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1 >> $out/pytest_gpu.txt
timeout 400 python bench.py --steps 50 --warmup 10 < /dev/null > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py < /dev/null > $out/bench_default_steps.json 2>> $out/bench.err
timeout 300 bash tools/gpu_profile.sh $tag --steps 5 --warmup 2 < /dev/null
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge ConformerCTCLarge EfficientConformerTransducerMedium; do
  timeout 300 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline < /dev/null > $out/${m}_bench.json 2> $out/${m}_bench.err
done
cat $out/pytest_gpu.txt
exit 0
