#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c3"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "stream or exact_mode_rejects" > "$out/t_stream.log" 2>&1; echo "stream tests rc=$?" | tee -a "$out/summary.txt"
tail -30 "$out/t_stream.log"
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_round3.py::test_streaming_and_causal_vs_reference_goldens > "$out/t_all.log" 2>&1; echo "all gpu tests rc=$?" | tee -a "$out/summary.txt"
tail -5 "$out/t_all.log"
timeout 300 python bench.py --ragged 1 --no-cpu-baseline --no-roofline > "$out/bench_ragged.json" 2>/dev/null; cut -c1-400 "$out/bench_ragged.json"
timeout 600 python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 > "$out/bench_fp32.json" 2> "$out/bench_fp32.err"; cut -c1-400 "$out/bench_fp32.json"; tail -2 "$out/bench_fp32.err"
