#!/bin/bash
# round 4, call 17: SQ counters of sx_gemm_kernel alone (where do its waves spend their cycles?)
set -u
out=$(pwd)/gpurun_out/r4_17; mkdir -p $out
repo=$(pwd)
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_g && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_g -o run -- python "$repo/tools/sx_gemm_bench.py" > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_g -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "python tools/sx_gemm_bench.py" > /dev/null
cat $out/sq_counters.txt | cut -c1-220
SQ2="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_h && rocprofv3 --kernel-trace --pmc $SQ2 -d /tmp/sq_h -o run -- python "$repo/tools/sx_gemm_bench.py" > "$out/sq2.log" 2>&1 )
db=$(find /tmp/sq_h -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters2.txt" "python tools/sx_gemm_bench.py" > /dev/null
cat $out/sq_counters2.txt | cut -c1-220; tail -3 $out/sq2.log
