#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_evidence_delta.sh <tag>
# The part of tools/gpu_evidence.sh that a change outside the default bf16 kernels invalidates: the default line and its kernel stats
# (to show they did not move), the label-exact mode, the Transducer configuration with its decode leg, the decode paths side by side,
# and the whole -m gpu suite.  Outputs under gpurun_out/<tag>/.
set -u
tag=$1
repo=$(pwd)
out="$repo/gpurun_out/$tag"
mkdir -p "$out"
python bench.py --steps 50 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"
tools/gpu_profile.sh "$tag" --steps 5 --warmup 2
python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 > "$out/bench_fp32.json" 2>> "$out/bench.err"
python bench.py --model EfficientConformerTransducerMedium --steps 5 --warmup 2 --no-cpu-baseline > "$out/EfficientConformerTransducerMedium_bench.json" 2>> "$out/bench.err"
( python tools/rnnt_diag.py 0.0 256; python tools/rnnt_diag.py 1.2 96; python tools/rnnt_diag.py 3.0 40 ) 2>&1 | grep -v amdgpu.ids > "$out/rnnt_decode_paths.txt"
python -m pytest tests -m gpu -q 2>&1 | tail -5 > "$out/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> "$out/pytest_gpu.txt"
ls -la "$out"; cat "$out/pytest_gpu.txt"; head -c 400 "$out/bench.json"
