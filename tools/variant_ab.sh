#!/bin/bash
# tools/variant_ab.sh <tag> [workload]: bench stage times per NL_SWEEP_VARIANT (3 = masks, 5/6 = half-shell), interleaved
TAG=$1; WL=${2:-cfg2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2; do for v in ${VARIANTS:-3 5 6}; do
  NL_SWEEP_VARIANT=$v timeout -k 10 300 python bench.py --steps 50 --warmup 5 --workload $WL --no-cpu-baseline --no-cfg4-baseline > $OUT/bench_${WL}_v${v}_$r.log 2>&1
  python - $OUT/bench_${WL}_v${v}_$r.log "$WL variant $v" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED", open(sys.argv[1]).read()[-800:]); sys.exit(0)
d = json.loads(l[-1]); st = d["roofline"]["stages_ms"]
print(f"{sys.argv[2]:18s} {d['ms_per_step']:.4f} ms/build  count {st['count']:.4f}  fill {st['fill']:.4f}  pairs {d['config'].get('half_pairs_reference')} checksum {d['config'].get('list_checksum_reference')}")
PY
done; done | tee -a $OUT/summary.txt
