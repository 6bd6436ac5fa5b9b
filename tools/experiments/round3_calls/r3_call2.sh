#!/bin/bash
set -u
repo=$(pwd); out="$repo/gpurun_out/r3c2"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "ragged" > "$out/t_ragged.log" 2>&1; echo "ragged tests rc=$?" | tee -a "$out/summary.txt"
tail -25 "$out/t_ragged.log"
for r in 1 0; do
  timeout 600 python bench.py --ragged $r --no-cpu-baseline > "$out/bench_ragged$r.json" 2> "$out/bench_ragged$r.err"; echo "bench ragged=$r rc=$?" | tee -a "$out/summary.txt"
  tail -3 "$out/bench_ragged$r.err"
done
python - <<'PY' | tee -a "$out/summary.txt"
import json
for r in (1,0):
    try:
        j=json.load(open("gpurun_out/r3c2/bench_ragged%d.json"%r))
        print("ragged=%d value %.3fM ms %.3f padded %.3fM check %s" % (r, j["value"]/1e6, j["ms_per_step"], j["config"]["padded_frames_per_s"]/1e6, json.dumps({k:v for k,v in (j.get("check") or {}).items() if k!="note"})))
        print("   classes", {k: round(v["ms_per_step"],3) for k,v in j.get("kernel_classes",{}).items()}, "roofline", round(j["roofline"]["frac"],4))
    except Exception as e: print("parse failed", r, e)
PY
for s in 2 4; do timeout 300 python bench.py --ragged 1 --streams $s --no-cpu-baseline --no-roofline --no-check 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ragged streams=$s', round(j['value']/1e6,2), round(j['ms_per_step'],3))" | tee -a "$out/summary.txt"; done
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ovl && timeout 600 rocprofv3 --kernel-trace -d /tmp/ovl -o run -- python "$repo/tools/overlap_probe.py" --steps 3 > "$out/overlap_probe.log" 2>&1 )
db=$(find /tmp/ovl -name "*.db" | head -1)
python - "$db" > "$out/db_schema.txt" 2>&1 <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, typ in c.execute("select name, type from sqlite_master where type in ('table','view')"):
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % name)]
    print(typ, name, cols)
view = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith("kernels")][0]
rows = c.execute("select * from %s order by start limit 3" % view).fetchall()
for r in rows: print(r)
PY
head -40 "$out/db_schema.txt"
