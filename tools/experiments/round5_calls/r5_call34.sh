#!/bin/bash
# round 5, call 34: depthwise convolution on the matrix pipe (dwconv_mfma_kernel, default) against dwconv_kernel: tests, then the step both ways
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_34; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k "matrix_pipe" < /dev/null 2>&1 | tail -12 | tee $out/pytest_new.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "conv or ragged or golden or stream or module" < /dev/null 2>&1 | tail -6 | tee $out/pytest.txt
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
exit 0
