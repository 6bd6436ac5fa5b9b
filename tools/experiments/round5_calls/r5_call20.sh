#!/bin/bash
# round 5, call 20: --ragged 0 roofline leg on round 4's kernels (chain_pair = 0, chain_variant = 0): does gemm_other 14.6 / 17.1 ms reproduce?
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_20; mkdir -p $out
for extra in "--opt chain_pair=0 --opt chain_variant=0" "--opt chain_pair=0" "--opt chain_variant=0"; do
timeout 600 python bench.py --no-cpu-baseline --ragged 0 --steps 10 --warmup 3 $extra 2>>$out/err.txt | grep '^{' > $out/b.json
python - <<PY
import json
d=json.load(open('$out/b.json'))
print('$extra', 'ms_per_step', round(d['ms_per_step'],3), ' gemm_other', round(d['kernel_classes']['gemm_other']['ms_per_step'],3), d['kernel_classes']['gemm_other']['launches_per_step'], ' gemm_ffn', round(d['kernel_classes']['gemm_ffn']['ms_per_step'],3))
PY
done
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_r0 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_r0 -o run -- python "$repo/bench.py" --no-cpu-baseline --ragged 0 --steps 3 --warmup 1 --opt chain_pair=0 --opt chain_variant=0 > "$out/trace_r0.log" 2>&1 )
db=$(find /tmp/kt_r0 -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$db")
rows=c.execute("select name, count(*), sum(end-start)/1000.0, max(end-start)/1000.0 from kernels group by name order by 4 desc limit 8").fetchall()
for r in rows: print("%-90s n %5d total %10.1f us  max %9.1f us" % (r[0][:90], r[1], r[2], r[3]))
PY
exit 0
