#!/bin/bash
# L2 hit / miss of the RNN-T cluster decode (B = 256), one workgroup -> XCD mapping per run
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c27.log; : > $L
cd /tmp && export TMPDIR=/tmp
for mode in "1:0" "1:1"; do
  tag=m$(echo $mode | tr ':' '_')
  rm -rf /tmp/pm_$tag
  RNNT_MODES=$mode timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d /tmp/pm_$tag -o run -- python $repo/tools/rnnt_diag.py 0.0 256 > /tmp/pm_$tag.log 2>&1
  db=$(find /tmp/pm_$tag -name "*.db" | head -1)
  echo "== cluster:by_slice = $mode" >> $L
  grep "^B " /tmp/pm_$tag.log >> $L
  python $repo/tools/sq_summary.py "$db" /tmp/pm_$tag.txt x | grep -E "^kernel|rnnt" | cut -c1-40,70-200 >> $L
done
cat $L
