#!/bin/bash
# tools/gpu_session.sh -- one gpurun call: smoke, GPU tests, bench, rocprofv3 passes.
# Every step runs under its own timeout; a step that times out or is killed ends the session (no further GPU
# work is started after a hang).  Outputs go to gpurun_out/<tag>/.
#   usage: tools/gpu_session.sh <tag> [steps...]     steps: smoke micro tests bench prof pmc
set -u
TAG=${1:-s}
shift || true
STEPS=${*:-"smoke micro tests bench prof pmc"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp

run() {  # run <seconds> <logfile> cmd...
  local secs=$1 log=$2
  shift 2
  echo "== $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "TIMEOUT/KILL in: $* -- stopping the session here"
    tail -5 "$log"
    exit $rc
  fi
  return $rc
}

for s in $STEPS; do
  case $s in
    smoke) run 300 "$OUT/smoke.log" python -c "import __graft_entry__ as g; g.smoke()"; tail -3 "$OUT/smoke.log" ;;
    fillab) run 300 "$OUT/fill_ab.log" python tools/fill_ab.py; cat "$OUT/fill_ab.log" ;;
    bwprobe) run 300 "$OUT/bw_probe.log" python tools/bw_probe.py; cat "$OUT/bw_probe.log" ;;
    dbgpar) run 300 "$OUT/debug_parity.log" python tools/debug_parity.py; cat "$OUT/debug_parity.log" ;;
    stages) run 300 "$OUT/stages.log" python tools/stage_times.py; cat "$OUT/stages.log" ;;
    micro) run 120 "$OUT/microbench.log" ./tools/microbench; cat "$OUT/microbench.log" ;;
    microtail) export MICRO_TAIL=1; run 120 "$OUT/microbench_tail.log" ./tools/microbench; cat "$OUT/microbench_tail.log" ;;
    tests) run 1100 "$OUT/pytest_gpu.log" python -m pytest tests -m gpu -q -x --durations=12; tail -25 "$OUT/pytest_gpu.log" ;;
    tests_k) run 900 "$OUT/pytest_gpu_k.log" python -m pytest tests -m gpu -q -x --durations=8 -k "$TESTS_K"; tail -25 "$OUT/pytest_gpu_k.log" ;;
    bench) run 420 "$OUT/bench.log" python bench.py --steps 50 --warmup 5; tail -3 "$OUT/bench.log" ;;
    bench_wl) run 420 "$OUT/bench_${WL}.log" python bench.py --steps ${WL_STEPS:-50} --warmup 5 --workload $WL --no-cpu-baseline; tail -2 "$OUT/bench_${WL}.log" ;;
    bench_wl_cpu) run 600 "$OUT/bench_${WL}.log" python bench.py --steps ${WL_STEPS:-50} --warmup 5 --workload $WL; tail -2 "$OUT/bench_${WL}.log" ;;
    prof)  # WL=cfg3 (etc.) profiles another workload; outputs are suffixed with it
      W=${WL:-cfg2}; rm -rf "$OUT/prof_$W"
      run 420 "$OUT/rocprof_stats_$W.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$W" -- python bench.py --steps ${WL_STEPS:-20} --warmup 3 --no-cpu-baseline --no-cfg4-baseline --workload $W
      find "$OUT/prof_$W" -name "*kernel_stats.csv" -exec cat {} \; | head -30 ;;
    pmc)
      W=${WL:-cfg2}
      rm -rf "$OUT/pmc_r" "$OUT/pmc_w" "$OUT/pmc_sq" "$OUT/pmc_sq2"
      PB="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cfg4-baseline --profile-reps 1 --workload $W"
      run 420 "$OUT/pmc_r.log" rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_r" -- $PB &&
      run 420 "$OUT/pmc_w.log" rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -- $PB &&
      run 420 "$OUT/pmc_sq.log" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $PB &&
      run 420 "$OUT/pmc_sq2.log" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LEVEL_WAVES --output-format csv -d "$OUT/pmc_sq2" -- $PB
      python tools/summarize_pmc.py "$OUT/pmc_r" "$OUT/pmc_w" "$OUT/pmc_sq" "$OUT/pmc_sq2" "$OUT/prof_$W" --json "$OUT/pmc_summary_$W.json" > "$OUT/pmc_summary_$W.txt" 2>&1; grep -A40 "k_sweep" "$OUT/pmc_summary_$W.txt" | head -90 ;;
    *) echo "unknown step $s" ;;
  esac
done
echo "session $TAG done"
