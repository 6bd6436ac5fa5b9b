repo=$(pwd); out="$repo/gpurun_out/r6_94_sq"; mkdir -p "$out"
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_f && rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_f -o run -- python "$repo/bench.py" --model EfficientConformerCTCLarge --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 > "$out/sq.log" 2>&1 )
db=$(find /tmp/sq_f -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters.txt" "bench.py --model EfficientConformerCTCLarge" > /dev/null
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_g && rocprofv3 --kernel-trace --pmc $SQ2 -d /tmp/sq_g -o run -- python "$repo/bench.py" --model EfficientConformerCTCLarge --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 > "$out/sq2.log" 2>&1 )
db=$(find /tmp/sq_g -name "*.db" | head -1)
python tools/sq_summary.py "$db" "$out/sq_counters2.txt" "bench.py --model EfficientConformerCTCLarge" > /dev/null
