#!/bin/bash
# tools/ab_rows.sh <tag> lib...: same-box A/B of builds of libnl_hip.so on cfg2 and cfg3 (stage times), plus the 27-cell path
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
run() {  # run <name> <lib> <NL_ROWS> <workload>
  NL_ROWS=$3 NL_HIP_LIB=$PWD/$2 timeout -k 10 200 python bench.py --steps ${STEPS:-100} --warmup 10 --workload $4 --no-cpu-baseline --no-cfg4-baseline > "$OUT/$1_$4.log" 2>&1 || { echo "FAILED $1 $4"; tail -5 "$OUT/$1_$4.log"; return; }
  python - "$OUT/$1_$4.log" "$1" "$4" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
st = d["roofline"]["stages_ms"]
print(f"{sys.argv[3]} {sys.argv[2]:16s}: {d['ms_per_step']:.4f} ms/build  count {st['count']:.4f}  fill {st['fill']:.4f}  reorder {st['reorder']:.4f}  pairs {d['config']['half_pairs_reference']} checksum {d['config']['list_checksum_reference']}", flush=True)
PY
}
for wl in cfg2 cfg3; do
  run cells27 "$1" 0 $wl
  for lib in "$@"; do run "$(basename $lib .so)" "$lib" -1 $wl; done
done | tee "$OUT/summary.txt"
