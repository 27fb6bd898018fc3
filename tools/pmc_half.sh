set -u
export TMPDIR=/tmp NL_SWEEP_VARIANT=5
OUT=gpurun_out/r2f; mkdir -p $OUT
PB="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cfg4-baseline --profile-reps 1"
for f in 0 1; do
  export NL_DEBUG_FLAGS=$f
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_$f -- $PB > $OUT/sq_$f.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2_$f -- $PB > $OUT/sq2_$f.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$f -- $PB > $OUT/prof_$f.log 2>&1
  python tools/summarize_pmc.py $OUT/sq_$f $OUT/sq2_$f $OUT/prof_$f --json $OUT/summary_$f.json > $OUT/summary_$f.txt 2>&1
  grep -A30 "k_sweep_half" $OUT/summary_$f.txt | head -40
done
