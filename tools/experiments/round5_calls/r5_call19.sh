#!/bin/bash
# round 5, call 19: the --ragged 0 roofline leg WITH the self-check in front of it (the run that showed gemm_other 14.6 / 17.1 ms in round 4)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_19; mkdir -p $out
timeout 600 python bench.py --no-cpu-baseline --ragged 0 --steps 10 --warmup 3 2>$out/err.txt | grep '^{' > $out/bench_ragged0_check.json
python - <<PY
import json
d=json.load(open('$out/bench_ragged0_check.json'))
print('ragged0+check ms_per_step', d['ms_per_step'])
for k,v in d['kernel_classes'].items(): print('  ', k, round(v['ms_per_step'],3), v['launches_per_step'])
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_r0 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_r0 -o run -- python "$repo/bench.py" --no-cpu-baseline --ragged 0 --steps 3 --warmup 1 > "$out/trace_r0.log" 2>&1 )
db=$(find /tmp/kt_r0 -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$db")
rows=c.execute("select name, count(*), sum(end-start)/1000.0, max(end-start)/1000.0 from kernels group by name order by 4 desc limit 14").fetchall()
for r in rows: print("%-90s n %5d total %10.1f us  max %9.1f us" % (r[0][:90], r[1], r[2], r[3]))
PY
exit 0
