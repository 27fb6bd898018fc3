#!/bin/bash
# tools/ab_libs.sh -- same-box A/B of several builds of libnl_hip.so (NL_HIP_LIB): each library runs the cfg2 bench
# ROUNDS times, interleaved, so that box-to-box and warm-up differences cancel.  usage: tools/ab_libs.sh <tag> lib...
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
WL=${WL:-cfg2}
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    NL_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps ${STEPS:-100} --warmup 10 --workload $WL --no-cpu-baseline --no-cfg4-baseline > "$OUT/${name}_$r.log" 2>&1 || { echo "FAILED $lib"; tail -5 "$OUT/${name}_$r.log"; exit 1; }
    python - "$OUT/${name}_$r.log" "$name" "$r" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
st = d["roofline"]["stages_ms"]
print(f"{sys.argv[2]:28s} round {sys.argv[3]}: {d['ms_per_step']:.4f} ms/build  count {st['count']:.4f}  fill {st['fill']:.4f}  reorder {st['reorder']:.4f}  pairs {d['config']['half_pairs_reference']} checksum {d['config']['list_checksum_reference']}")
PY
  done
done | tee "$OUT/summary.txt"
