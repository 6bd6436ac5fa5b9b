#!/bin/bash
# round 5, call 36: kernel durations of the two depthwise-convolution kernels (kernel trace of the default step and of a one-stream step)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_36; mkdir -p $out
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
