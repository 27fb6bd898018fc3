#!/bin/bash
# tools/pmc_rows.sh <tag> <workload> <NL_ROWS value>: SQ counter passes + kernel stats of bench.py (instruction mix only)
set -u
export TMPDIR=/tmp NL_ROWS=$3
OUT=gpurun_out/$1/${2}_rows$3; W=$2; mkdir -p $OUT
PB="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cfg4-baseline --profile-reps 1 --workload $W"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -- $PB > $OUT/sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -- $PB > $OUT/sq2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $PB > $OUT/prof.log 2>&1
python tools/summarize_pmc.py $OUT/sq $OUT/sq2 $OUT/prof --json $OUT/summary.json > $OUT/summary.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-120 | head -12
