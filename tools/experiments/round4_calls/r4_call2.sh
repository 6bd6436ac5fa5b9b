#!/bin/bash
# round 4, call 2: the split-precision mode (csrc/split.hip) - parity tests of both label-exact modes, the rest of the suite after call 1's
# first failure (streaming tolerance), bench.py --precision split
set -u
out=gpurun_out/r4_02; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $out/pytest_gpu.txt
timeout 600 python bench.py --precision split --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_split.json 2> $out/bench_split.err
cat $out/pytest_gpu.txt; head -c 1500 $out/bench_split.json; echo; tail -5 $out/bench_split.err
