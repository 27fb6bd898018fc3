#!/bin/bash
# tools/ab_env.sh -- same-box A/B of one environment switch of the library: bench.py alternately with VAR=a and VAR=b.
#   usage: [WL=cfg3] [ROUNDS=2] [STEPS=50] tools/ab_env.sh <tag> VAR a b [c ...]
set -u
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
WL=${WL:-cfg2}
for r in $(seq 1 ${ROUNDS:-2}); do
  for val in "$@"; do
    env $VAR=$val timeout -k 10 300 python bench.py --steps ${STEPS:-50} --warmup 10 --workload $WL --no-cpu-baseline --no-cfg4-baseline > "$OUT/${VAR}_${val}_$r.log" 2>&1 || { echo "FAILED $VAR=$val"; tail -5 "$OUT/${VAR}_${val}_$r.log"; exit 1; }
    python - "$OUT/${VAR}_${val}_$r.log" "$WL $VAR=$val" "$r" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
st = d["roofline"]["stages_ms"]
print(f"{sys.argv[2]:28s} round {sys.argv[3]}: {d['ms_per_step']:.4f} ms/build  count {st['count']:.4f}  fill {st['fill']:.4f}  reorder {st['reorder']:.4f}  pairs {d['config']['half_pairs_reference']} checksum {d['config']['list_checksum_reference']}")
PY
  done
done | tee "$OUT/summary.txt"
