#!/bin/bash
# tools/pipe_ab.sh <tag> [workload]: bench stage times with the pipelined COUNT sweep off / on (NL_PIPE), interleaved
TAG=$1; WL=${2:-cfg2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2; do for p in ${PIPES:-0 4}; do
  NL_PIPE=$p timeout -k 10 300 python bench.py --steps 50 --warmup 5 --workload $WL --no-cpu-baseline --no-cfg4-baseline > $OUT/bench_${WL}_p${p}_$r.log 2>&1
  python - $OUT/bench_${WL}_p${p}_$r.log "$WL NL_PIPE=$p" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED", open(sys.argv[1]).read()[-800:]); sys.exit(0)
d = json.loads(l[-1]); st = d["roofline"]["stages_ms"]
print(f"{sys.argv[2]:18s} {d['ms_per_step']:.4f} ms/build  count {st['count']:.4f}  fill {st['fill']:.4f}  pairs {d['config'].get('half_pairs_reference')} checksum {d['config'].get('list_checksum_reference')}")
PY
done; done | tee -a $OUT/summary.txt
