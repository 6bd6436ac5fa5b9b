#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c5"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "ragged" > "$out/t_ragged.log" 2>&1; echo "ragged tests rc=$?" | tee -a "$out/summary.txt"
tail -12 "$out/t_ragged.log"
timeout 600 python bench.py --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge EfficientConformerTransducerMedium ConformerCTCLarge; do
  timeout 600 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > "$out/${m}_bench.json" 2> "$out/${m}_bench.err"; echo "$m rc=$?" | tee -a "$out/summary.txt"
done
python - <<'PY' | tee -a "$out/summary.txt"
import json, glob
for f in sorted(glob.glob("gpurun_out/r3c5/*bench.json")):
    try:
        j=json.load(open(f))
        print(f.split("/")[-1], "value %.3fM ms %.3f" % (j["value"]/1e6, j["ms_per_step"]), "check", (j.get("check") or {}).get("ok"), "frac", round(j["roofline"]["frac"],4), j["roofline"]["kernel"][:30], j.get("transducer_legs"))
        print("    ", {k: round(v["ms_per_step"],3) for k,v in j.get("kernel_classes",{}).items()})
    except Exception as e: print(f, "parse failed", e)
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ovl && timeout 600 rocprofv3 --kernel-trace -d /tmp/ovl -o run -- python "$repo/tools/overlap_probe.py" --steps 6 > "$out/overlap_probe.log" 2>&1 )
db=$(find /tmp/ovl -name "*.db" | head -1)
python tools/overlap_timeline.py "$db" "$out/overlap_timeline.txt"; cat "$out/overlap_timeline.txt"
tools/gpu_profile.sh r3c5prof --steps 5 --warmup 2 ; head -30 gpurun_out/r3c5prof/kernel_stats.txt
