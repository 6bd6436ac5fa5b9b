#!/bin/bash
# round 5, call 37: dwconv_mfma_kernel with 16-byte staging loads / stores (kernel size 15 by default): tests, step A / B, kernel durations
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_37; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "matrix_pipe or conv or ragged or golden or stream or module" < /dev/null 2>&1 | tail -6 | tee $out/pytest.txt
bench() {
  tag=$1; shift
  for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-check "$@" < /dev/null 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4))" | tee -a $out/ab.txt; done
}
bench mfma
bench valu --opt dwconv_mfma=0
bench mfma
bench valu --opt dwconv_mfma=0
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  for shape in "--streams 3" "--streams 1 --ranges 1"; do
    tag="mfma${v}_$(echo $shape | tr -d ' -')"
    rm -rf /tmp/prof_$tag
    timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python $repo/bench.py --no-cpu-baseline --no-roofline --no-check --steps 5 --warmup 2 $shape --opt dwconv_mfma=$v < /dev/null > $out/prof_$tag.log 2>&1
    db=$(find /tmp/prof_$tag -name "*.db" | head -1)
    if [ -n "$db" ]; then
      python $repo/tools/rocprof_summary.py "$db" $out/stats_$tag.txt "bench.py $shape --opt dwconv_mfma=$v" < /dev/null > /dev/null 2>&1
      echo "== dwconv_mfma=$v $shape" >> $out/dw_kernels.txt
      grep -i "dwconv" $out/stats_$tag.txt | cut -c1-60,100-200 >> $out/dw_kernels.txt
    fi
  done
done
cat $out/dw_kernels.txt
exit 0
