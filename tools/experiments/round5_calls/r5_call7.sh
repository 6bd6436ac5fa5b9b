#!/bin/bash
# round 5, call 7: what a kernel family costs the STEP under the three-stream overlap (ablation library: launches of a family dropped, timing only)
set -u
repo=$(pwd); out=$repo/gpurun_out/r5_07; mkdir -p $out
export EFFCONF_ABLATE_LIB=$repo/efficientconformer_amd/build/libeffconf_ablate.so
run() {
  tag=$1; mask=$2; shift; shift
  EFFCONF_SKIP=$mask timeout 300 python tools/ablate_bench.py --no-cpu-baseline --no-roofline --no-check "$@" 2>&1 | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-28s skip %3d  %.4f ms' % ('$tag', $mask, d['ms_per_step']))" | tee -a $out/ablate.txt
}
for rep in 1 2; do
run all 0
run no_attention 1
run no_chainA 2
run no_chainB 4
run no_dwconv 8
run no_mel 16
run no_subsample 32
run no_glue 64
run no_chains 6
run no_chains_attention 7
run only_attention 126
run only_chainA 125
done
echo "# one stream, one range" | tee -a $out/ablate.txt
run all_1s 0 --streams 1 --ranges 1
run no_attention_1s 1 --streams 1 --ranges 1
run no_chainA_1s 2 --streams 1 --ranges 1
exit 0
