#!/bin/bash
# LDS / VMEM counters of the chain kernels (one stream), and the list of available counters
repo=$(pwd); mkdir -p gpurun_out; L=$repo/gpurun_out/c18.log; : > $L
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]*LDS[A-Z0-9_]*|SQ_INSTS_[A-Z_0-9]*|SQ_INST_CYCLES_[A-Z_]*|SQ_WAIT_INST_[A-Z]*|SQ_ACTIVE_INST_[A-Z]*|TCP_[A-Z_0-9]*|TA_[A-Z_0-9]*BUSY[A-Z_0-9]*|SQ_VMEM[A-Z_0-9]*|SQ_IFETCH[A-Z_]*|SQ_WAVES[A-Z_]*)\b" | sort -u | tr '\n' ' ' >> $L
echo >> $L
run() { # name counters...
  name=$1; shift
  rm -rf /tmp/pm_$name
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pm_$name -o run -- python $repo/bench.py --no-cpu-baseline --no-roofline --no-check --steps 2 --warmup 1 --streams 1 > /tmp/pm_$name.log 2>&1
  db=$(find /tmp/pm_$name -name "*.db" | head -1)
  echo "== $name: $*" >> $L
  if [ -n "$db" ]; then python $repo/tools/sq_summary.py "$db" /tmp/pm_$name.txt x | grep -E "^kernel|chain_kernel" | cut -c1-36,70-240 >> $L; else tail -5 /tmp/pm_$name.log >> $L; fi
}
run lds SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run vm SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC
run w SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
cat $L
