#!/bin/bash
# refresh of the default-path evidence after the split-bf16 CTC head became the default
set -u
tag=r3_03; repo=$(pwd); out="$repo/gpurun_out/$tag"; mkdir -p "$out"
python bench.py --steps 50 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"
tools/gpu_profile.sh "$tag" --steps 5 --warmup 2
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge ConformerCTCLarge; do
  python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > "$out/${m}_bench.json" 2>> "$out/bench.err"
done
python -m pytest tests -m gpu -q 2>&1 | tail -4 > "$out/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> "$out/pytest_gpu.txt"
cat "$out/pytest_gpu.txt"; python - <<PY
import json
for f in ("bench", "EfficientConformerCTCMedium_bench", "EfficientConformerCTCLarge_bench", "ConformerCTCLarge_bench"):
    d = json.load(open("$out/" + f + ".json"))
    print(f, round(d["value"] / 1e6, 3), round(d["ms_per_step"], 3), d["check"]["ok"], d["check"].get("label_sequences_identical_to_oracle"), d["check"].get("argmax_flips_vs_oracle"))
PY
