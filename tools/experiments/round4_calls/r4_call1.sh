#!/bin/bash
# round 4, call 1: the whole -m gpu suite on the hygiene tree (new: split head vocab sizes, ragged sharded leg, bench check at N = 2),
# the default bench line with the two-shape roofline leg, the overlap probe in its four arrangements, the stream-count probe
set -u
out=gpurun_out/r4_01; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $out/pytest_gpu.txt
timeout 600 python bench.py --steps 30 --warmup 6 > $out/bench.json 2> $out/bench.err
for args in "--mode sync" "--mode pipelined" "--mode sync --range-frames 40,35,25" "--mode sync --range-frames 45,35,20" "--mode pipelined --world 2" "--no-collective"; do
  echo "## overlap_probe.py --wire bf16 $args" >> $out/overlap_events.txt
  timeout 300 python tools/overlap_probe.py --wire bf16 $args 2>/dev/null | grep -v amdgpu.ids >> $out/overlap_events.txt
done
timeout 600 python tools/stream_cliff_probe.py > $out/stream_cliff.txt 2>&1
cat $out/pytest_gpu.txt; head -c 600 $out/bench.json; echo; tail -3 $out/bench.err; cat $out/overlap_events.txt; cat $out/stream_cliff.txt
