#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_pmc.sh <tag> [bench.py args...]
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, as MI355X_MICROARCH.md prescribes) of bench.py
# (PMC_STREAMS = the --streams value of the profiled command, default 3 = bench.py's default; recorded in the json)
# -> gpurun_out/<tag>/pmc_hbm_traffic.txt and gpurun_out/<tag>/pmc_traffic.json (copy into profiles/ to have bench.py report it).
set -u
tag=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out/$tag"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_$c -o run -- python "$repo/bench.py" --no-cpu-baseline --no-roofline --no-check "$@" > "$repo/gpurun_out/$tag/pmc_$c.log" 2>&1
done
f=$(find /tmp/pmc_${tag}_FETCH_SIZE -name "*.db" | head -1)
w=$(find /tmp/pmc_${tag}_WRITE_SIZE -name "*.db" | head -1)
python "$repo/tools/pmc_summary.py" "$f" "$w" "$repo/gpurun_out/$tag/pmc_hbm_traffic.txt" "python bench.py --no-cpu-baseline --no-roofline --no-check $*" "$repo/gpurun_out/$tag/pmc_traffic.json" EfficientConformerCTCSmall 256 libri "${PMC_STREAMS:-3}" > /dev/null
